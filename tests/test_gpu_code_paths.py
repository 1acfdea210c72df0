"""The code-path matrix where the driver runs it (`pytest -m gpu`): the compact core of tests/code_path_core.py -- NTT round trip,
multiply -> relinearize -> rescale, a NAF rotation, BFV multiply + relinearize, one he355_rotate_sum level walk, one DotProduct through
the API-Bridge C ABI, every result held to the oracle bit for bit -- re-run under each setting that selects kernels or schedules.
Each setting runs in a fresh child process started BEFORE anything here touches the GPU for it (the switches are read at context
creation, some once per process); the whole module takes under a minute.  tools/test_matrix.sh runs the FULL suite under the same
settings (builder-run, ~15 minutes)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))

SETTINGS = {
    "defaults": {},
    "u64_engine_every_prime_shoup_form": {"HE355_FORCE_U64": "1"},
    "u64_engine_every_prime_shoup_form_no_ring_in_lds": {"HE355_FORCE_U64": "1", "HE355_LDS_MAX": "0"},
    "u64_engine_shoup_form_default_assignment": {"HE355_FORCE_U64": "shoup"},
    "behz_seal_61bit_base": {"HE355_BEHZ_BASE": "seal"},
    "behz_unfused_no_hoist": {"HE355_BEHZ_FUSE": "0"},
    # (the core's rings are N = 4096: without HE355_LDS_MAX=0 its small batches would all take the ring-in-LDS shape of round 6 and the
    # settings below would select nothing)
    "ring_in_lds_up_to_64": {"HE355_LDS_MAX": "64"},
    "no_ring_in_lds": {"HE355_LDS_MAX": "0"},
    "throughput_shape_for_every_batch": {"HE355_LATENCY_MAX": "0", "HE355_LDS_MAX": "0"},
    "latency_shape_up_to_64": {"HE355_LATENCY_MAX": "64", "HE355_LDS_MAX": "0"},
    "one_launch_per_engine": {"HE355_DUAL_ENGINE": "0", "HE355_LDS_MAX": "0"},
    "unfused_mod_down": {"HE355_K3_FUSE": "0", "HE355_LDS_MAX": "0"},
    "fused_everywhere_no_small_grid_rules": {"HE355_K3_FUSE": "all", "HE355_LDS_MAX": "0"},
    "node_by_node_walks": {"HE355_LEVEL_WALK": "0"},
    "device_pool_off": {"HE355_POOL": "0"},
    "single_stream_chunks_of_3": {"HE355_DUAL_STREAM": "0", "HE355_CHUNK": "3"},
}


@pytest.mark.parametrize("name", list(SETTINGS))
def test_core_under_setting(name):
    env = {k: v for k, v in os.environ.items() if not (k.startswith("HE355_") and k not in ("HE355_SEED", "HE355_LIB_PATH", "HE355_DEVICE"))}
    env.update(SETTINGS[name])
    r = subprocess.run([sys.executable, os.path.join(HERE, "code_path_core.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, f"{name} ({SETTINGS[name]}):\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    assert "code paths ok: ntt mul_relin_rescale rotate_naf bfv_multiply_relin rotate_sum bridge_dot" in r.stdout
