"""CPU: the accounting rules of bench.py that decide what the driver-visible line may claim (no GPU, no timing).

Each of them was once wrong in a committed line: an MLP tail timed as one launch but counted as two kernels, a
whole-path "fraction of the roof" above 1 on the inference grid, PMC traffic of the B = 8 workload attached to a
rank that ran 3 images."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

METRIC = "list_im2sdf_b8_n20k_224"


def test_fused_tail_moves_fc1_flops_into_the_one_launch_that_runs_them():
    B, N, img_res, vox_res, map_size, _ = bench.WORKLOADS[METRIC]
    two = bench.kernel_table(B, N, img_res, vox_res, map_size, 2, 2)
    one = bench.kernel_table(B, N, img_res, vox_res, map_size, 2, 2, fused_tail=True)
    assert one["fc_1"][1] == 0                                   # nothing is launched in that interval
    assert one["fc_2_out"][1] == two["fc_1"][1] + two["fc_2_out"][1]
    assert one["fc_0"] == two["fc_0"]
    # 2 (3610*512 + 512*256 + 256*256 + 256) FLOP per point over the three MLP entries, either way
    for t in (one, two):
        assert t["fc_0"][1] + t["fc_1"][1] + t["fc_2_out"][1] == B * N * bench.W_FLOP_PER_PT


def test_whole_path_fraction_is_physical_or_the_bench_refuses_to_print_it():
    # scattered queries (the metric): the no-reuse tap bytes bind in both modes (fp16 maps: 161 M points/s against an
    # MFMA roof of 611 M; fp32 maps of the bf16x3 mode: 80.8 M against 204 M at three products per MAC)
    r = bench.path_roofs(78.8e6, "fp16")
    assert r["binding"] == "hbm gather" and 0.4 < r["whole_path_frac_of_binding_roof"] < 0.6
    r = bench.path_roofs(44.4e6, "bf16x3")
    assert r["binding"] == "hbm gather" and r["mfma_products_per_mac"] == 3
    assert abs(r["gather_roof_points_per_s"] - 8e12 / 99056) < 1 and 0.5 < r["whole_path_frac_of_binding_roof"] < 0.6
    # the dense grid of one image runs ABOVE the no-reuse tap-byte figure (88 M > 80.8 M points/s): not a roof there
    g = bench.path_roofs(88.25e6, "bf16x3", grid=True)
    assert g["binding"].startswith("mfma") and "gather_roof_points_per_s" not in g
    assert g["no_reuse_tap_request_points_per_s"] < 88.25e6
    assert g["mfma_flop_per_point_executed"] == bench.W_FLOP_PER_PT - 2 * 1024 * 512
    assert 0.25 < g["whole_path_frac_of_binding_roof"] < 0.40
    with pytest.raises(AssertionError, match="not physical"):
        bench.path_roofs(88.25e6, "bf16x3")                      # the same rate against the scattered-query roofs
    with pytest.raises(AssertionError, match="not physical"):
        bench.path_roofs(1e9, "fp16", grid=True)


def test_pmc_traffic_belongs_to_the_batch_it_was_collected_on():
    B = bench.WORKLOADS[METRIC][0]
    assert bench.pmc_traffic("fp16", METRIC, 3) == {}            # strong scaling: this rank ran 3 images
    assert bench.pmc_traffic("fp16", METRIC, 64) == {}
    assert bench.pmc_traffic("fp16", "list_grid256_b1", 1) == {}
    got = bench.pmc_traffic("fp16", METRIC, B)
    # (empty as well when profiles/pmc_traffic.json is older than the kernel sources: stale counters are dropped)
    assert got == {} or "fc_0" in got


def test_profile_summaries_name_kernels_that_rocprof_left_mangled():
    """rocprofv3 does not demangle signatures with `_Float16*` parameters (PDF16_): the LDS-window and packed-half
    scatter kernels of the backward.  A summary keyed on demangled names filed them under "other: torch / rocclr"
    (60 ms of one trace) until round 3; both tools now read the mangled form as well."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import pmc_traffic
    import summarize_prof as sp
    assert sp.short('"_ZN4list17k_scatter_vox_winILi128ELi1ELi18432EEEvNS_13ScatterParamsE12ListVoxLeveliPDF16_"') \
        == "k_scatter_vox_win<128, 1, 18432>"
    assert sp.short("_ZN4list16k_scatter_vox_h2ILi32EEEvNS_13ScatterParamsE12ListVoxLeveliiPDF16_") == "k_scatter_vox_h2<32>"
    assert sp.short("_ZN4list13k_h16_to_gradEPKDF16_PflPKff") == "k_h16_to_grad"
    # the two spellings of one kernel agree, bools included
    assert sp.short("void list::k_gemm_nt_pp<0, 1, true, 0>(list::GemmParams)") == "k_gemm_nt_pp<0, 1, true, 0>"
    assert sp.short("_ZN4list12k_gemm_nt_ppILi0ELi1ELb1ELi0EEEvNS_10GemmParamsE") == "k_gemm_nt_pp<0, 1, true, 0>"
    # kernels of other libraries stay anonymous
    assert sp.short("void at::native::vectorized_elementwise_kernel<4, foo>(int)") is None
    assert sp.short("_ZN2at6native3fooEv") is None
    assert pmc_traffic.kernel_key("_ZN4list14k_mlp_tail_f16ENS_10TailParamsE", 320000, 0) == "fc_2_out"
    assert pmc_traffic.kernel_key("void list::k_gemm_nt_pp<0, 1, true, 0>(list::GemmParams)", 640000, 0) == "fc_0"

