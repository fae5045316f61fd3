"""Correctness of the launches that are ordered by something other than the stream's queue barrier
(hipExtAnyOrderLaunch behind the first launch of a group: the seven gathers of a chunk, the weight repacks): the
default dispatch against LIST_LAUNCH_IN_ORDER=1, each in a fresh process, bit for bit -- a multi-chunk sorted forward
twenty times back to back, an unsorted forward, the packed weight buffers, and queries right behind in-place weight
updates.  One run each (hip_ext.h documents the flag only as "can be launched in any order": run once per runtime
upgrade)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run_child(tmp_path, tag, env_extra):
    out = os.path.join(tmp_path, f"launch_order_{tag}.npz")
    env = dict(os.environ)
    env.pop("LIST_LAUNCH_IN_ORDER", None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_child_launch_order.py"), out], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return np.load(out)


def test_any_order_dispatch_equals_in_order_dispatch(tmp_path):
    a = _run_child(str(tmp_path), "default", {})
    b = _run_child(str(tmp_path), "in_order", {"LIST_LAUNCH_IN_ORDER": "1"})
    assert sorted(a.files) == sorted(b.files) and len(a.files) == 10
    for k in a.files:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
