"""Command-line configuration with the reference's flag names (arguments.py:4-133), plus the
flags of the MI355X path.  Unlike the reference, a missing test-list file is tolerated."""
import os
from argparse import ArgumentParser


def _bool(v):
    return str(v).lower() not in ("0", "false", "no", "off", "")


def build_parser():
    p = ArgumentParser(description="Image_to_3D (LIST) -- MI355X-native query path")
    p.add_argument("--cuda", type=_bool, default=True)
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("--plot_every_batch", type=int, default=10)
    p.add_argument("--save_every_epoch", type=int, default=25)
    p.add_argument("--save_after_epoch", type=int, default=1)
    p.add_argument("--test_every_epoch", type=int, default=25)
    p.add_argument("--load_pretrain", type=_bool, default=True)
    p.add_argument("--skip_train", action="store_true")
    p.add_argument("--viewnum", type=int, default=36)
    p.add_argument("--img_res", type=int, default=224)
    p.add_argument("--mcube_znum", type=int, default=128,
                   help="resolution of the inference query grid (decoupled from --vox_res)")
    p.add_argument("--test_pointnum", type=int, default=65536)
    p.add_argument("--chunk_s", type=int, default=0)
    p.add_argument("--chunk_l", type=int, default=217)
    p.add_argument("--chunk_id", type=int, default=0)
    p.add_argument("--chunk_num", type=int, default=4)
    p.add_argument("--model", type=str, help="e.g. network.models.LIST")
    p.add_argument("--dataset", type=str, help="e.g. datasets.Datasets.SyntheticIM2SDF")
    p.add_argument("--random_h_flip", action="store_true")
    p.add_argument("--color_jitter", action="store_true")
    p.add_argument("--normalize", action="store_true")
    p.add_argument("--point_decoder", action="store_true")
    p.add_argument("--warm_start", action="store_true")
    p.add_argument("--lr", type=float, default=0.0001)
    p.add_argument("--beta1", type=float, default=0.9)
    p.add_argument("--cam_batch_size", type=int, default=16)
    p.add_argument("--cam_lr", type=float, default=0.00005)
    p.add_argument("--train_batch_size", type=int, default=12)
    p.add_argument("--test_batch_size", type=int, default=1)
    p.add_argument("--epochs", type=int, default=300)
    p.add_argument("--sampling_mode", type=str, default="weighted")
    p.add_argument("--exp_name", "-e", type=str, default="d2im+tGCN")
    p.add_argument("--eval_pred", action="store_true")
    p.add_argument("--supervise_proj", action="store_true")
    p.add_argument("--coarse_point_density", type=int, default=10000)
    p.add_argument("--sample_point_density", type=int, default=32768)
    p.add_argument("--sdf_max_dist", type=float, default=1.0)
    p.add_argument("--sdf_scale", type=float, default=1.0)
    p.add_argument("--weight_decay", type=float, default=1e-5)
    p.add_argument("--sigmas", type=float, nargs="+", default=[0.003, 0.01, 0.07])
    p.add_argument("--sample_distribution", type=float, nargs="+", default=[0.5, 0.49, 0.01])
    p.add_argument("--point_feat", type=int, nargs="+", default=[128, 128, 256, 256, 256, 128, 128, 3])
    p.add_argument("--point_degree", type=int, nargs="+", default=[2, 2, 2, 2, 2, 2, 64])
    p.add_argument("--im_enc_layers", type=int, nargs="+", default=[1, 1, 1, 1, 16, 32, 64, 128, 128])
    p.add_argument("--n_decoder_pos", type=int, default=2)
    p.add_argument("--bb_min", type=float, default=-0.5)
    p.add_argument("--bb_max", type=float, default=0.5)
    p.add_argument("--vox_res", type=int, default=128)
    p.add_argument("--data_dir", default="./Datasets/shapenet/")
    p.add_argument("--mesh_dir", default="./Datasets/shapenet/mesh/")
    p.add_argument("--h5_dir", default="./Datasets/shapenet/sampled_points/")
    p.add_argument("--cam_dir", default="./Datasets/shapenet/images/")
    p.add_argument("--image_dir", default="./Datasets/shapenet/images/")
    p.add_argument("--split_dir", default="./data/DISN_split/", help="{cat}_{train,test}.lst shape-id lists")
    p.add_argument("--catlist", type=str, nargs="+",
                   default=["03001627", "02691156", "02828884", "02933112", "03211117", "03636649",
                            "03691459", "04090263", "04256520", "04379243", "04530566", "02958343",
                            "04401088"])
    p.add_argument("--output_dir", default="./results/")
    p.add_argument("--test_cam_id", type=int, default=2)
    p.add_argument("--test_gpu_id", type=int, default=0)
    p.add_argument("--test_checkpoint", default="best_model_test.pt.tar")
    p.add_argument("--testlist_file", default="./data/DISN_split/testlist_all.lst")
    # --- MI355X path
    p.add_argument("--precision", default="bf16x3", choices=["bf16x3", "fp16", "bf16"],
                   help="arithmetic of the implicit MLP on the HIP path")
    p.add_argument("--vox_encoder_precision", default="fp32", choices=["fp32", "fp16"],
                   help="fp16: the 3-D encoder runs under autocast (MIOpen half kernels, 2.5x faster at 128^3) and "
                        "hands fp16 channels-last levels to the query path, which uses them where they lie; "
                        "changes the features by ~3e-4, so it is opt-in and meant for --precision fp16")
    p.add_argument("--channels_last", type=_bool, default=True,
                   help="run the encoders that feed the query path in channels-last memory format")
    p.add_argument("--synthetic_len", type=int, default=64, help="items per epoch of the synthetic datasets")
    p.add_argument("--num_workers", type=int, default=0)
    p.add_argument("--max_steps", type=int, default=0, help="stop training after this many batches (0 = off)")
    return p


def finalize(args):
    testlist = []
    if args.testlist_file and os.path.exists(args.testlist_file):
        with open(args.testlist_file) as f:
            for line in f.readlines()[:30]:
                parts = line.strip().split(" ")
                if len(parts) >= 3 and parts[0] in args.catlist:
                    testlist.append({"cat_id": parts[0], "shape_id": parts[1], "cam_id": parts[2]})
    args.testlist = testlist
    args.checkpoint_dir = args.output_dir + args.exp_name + "/checkpoints/"
    args.results_dir = args.output_dir + args.exp_name + "/"
    args.log = args.output_dir + args.exp_name + "/log.txt"
    return args


def get_args(argv=None):
    return finalize(build_parser().parse_args(argv))


def default_config(**overrides):
    """Config namespace with the defaults, for programmatic use (tests, bench)."""
    args = build_parser().parse_args([])
    for k, v in overrides.items():
        setattr(args, k, v)
    return finalize(args)
