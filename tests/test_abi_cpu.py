"""The C-ABI shared library: builds for gfx950, loads without a GPU, exports every symbol that
``include/orl_hip.h`` declares, validates arguments before touching the device.  CPU only."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from openrl_amd import _native

    return _native.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "orl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(orl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = _declared_symbols()
    for must in ("orl_gae_scan", "orl_adv_normalize_pack", "orl_buffer_insert", "orl_gather_minibatch", "orl_act_step",
                 "orl_ppo_fwd_bwd", "orl_ppo_reduce", "orl_ppo_apply", "orl_valuenorm_update", "orl_rollout_fused",
                 "orl_env_step", "orl_env_reset", "orl_perm_feistel", "orl_version", "orl_last_error_string"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    from openrl_amd import _native

    declared = _declared_symbols()
    bound = set(_native.exported_symbols())
    for name in declared:
        assert hasattr(lib, name), "liborl_hip.so does not export %s" % name
        assert name in bound, "openrl_amd/_native.py does not bind %s" % name
    assert bound <= set(declared), "bindings without a declaration: %s" % (bound - set(declared))


def test_host_side_queries(lib):
    from openrl_amd import _native as n

    assert lib.orl_version() == n.ORL_VERSION == 306
    pol = n.NetDesc(4, 64, 2, n.ORL_HEAD_CATEGORICAL)
    cri = n.NetDesc(4, 64, 1, n.ORL_HEAD_VALUE)
    gau = n.NetDesc(17, 64, 6, n.ORL_HEAD_GAUSSIAN)
    # parameter counts of the reference towers at the CartPole shape (4 866 / 4 804)
    assert lib.orl_param_count(C.byref(pol)) == 4866
    assert lib.orl_param_count(C.byref(cri)) == 4801 + 0  # 4 804 in the reference includes 3 ValueNorm scalars
    assert lib.orl_param_count(C.byref(gau)) == 17 * 64 + 64 * 3 + 64 * 64 + 64 * 3 + 6 * 64 + 6 + 6
    assert lib.orl_record_width(4, 4, 1, 2) == 16  # one 64-byte record per sample at config 2
    assert lib.orl_record_width(17, 17, 6, 0) == 52
    assert lib.orl_ppo_max_blocks() == 256
    assert lib.orl_gae_max_partials(128, 4096) == 256  # 16 lanes per workgroup
    assert lib.orl_env_state_width(n.ORL_ENV_SYNTH) == 4 and lib.orl_env_state_width(n.ORL_ENV_CARTPOLE) == 8


def test_argument_validation_happens_before_any_launch(lib):
    """Bad arguments return ORL_E_INVALID / ORL_E_UNSUPPORTED with a message - no GPU is touched."""
    from openrl_amd import _native as n

    rc = lib.orl_gae_scan(None, None, None, None, None, None, None, 4, 4, 0.99, 0.95, 1, None, None, None, None, None)
    assert rc == -1 and b"null" in lib.orl_last_error_string()
    rc = lib.orl_perm_feistel(None, 0, 0, 0, None)
    assert rc == -1
    bad = n.NetDesc(4, 128, 2, n.ORL_HEAD_CATEGORICAL)  # hidden_size 128 is not built
    rc = lib.orl_act_step(C.byref(bad), None, None, None, None, None, None, 1, 0, 0, 0, 0, None, None, None, None, None, None)
    assert rc == -2 and b"hidden_size" in lib.orl_last_error_string()
    assert lib.orl_env_state_width(77) == -1


def test_no_cpu_fallback_in_the_product_path():
    """The engine refuses to run without a HIP device instead of silently computing on the CPU."""
    import torch

    from openrl_amd import _native as n
    from openrl_amd import ops

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(n.NativeError):
        n.require_gpu("cpu")
    with pytest.raises(n.NativeError):
        ops.perm_feistel(16, 0, 0, "cuda:0")
    from openrl_amd.envs.common import make

    with pytest.raises(n.NativeError):
        make("CartPole-v1", env_num=4)


def test_product_package_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "openrl_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_header_is_plain_c():
    """include/orl_hip.h is the C ABI: it must compile as C99 on its own (no C++ / torch / HIP types in the signatures)."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc on this machine")
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "orl_hip.h")
    res = subprocess.run([gcc, "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", hdr], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_fused_general_tower_host_queries(lib):
    """orl_gt_supported / orl_gt_image_floats / orl_gt_raw_floats (no launch): which general towers the cross-layer fused
    kernels take, and the sizes a caller allocates (csrc/orl_gen_tower.h: GtLay)."""
    from openrl_amd import _native as n

    def desc(H, D, n_layers, heads=(2,)):
        d = n.GtDesc()
        d.theta, d.D, d.H, d.n_layers, d.n_heads = 1, D, H, n_layers, len(heads)
        d.o_fn_g = d.o_fn_be = -1
        for k in range(min(n_layers, n.ORL_GT_MAX_LAYERS)):
            d.act[k] = n.ORL_ACT_RELU if k + 1 < n_layers else n.ORL_ACT_NONE
        for k, h in enumerate(heads):
            d.head_n[k] = h
        return d

    d = desc(128, 4, 2)
    assert lib.orl_gt_supported(C.byref(d)) == 1
    # raw sums: G0 [128 x 16] + G1 [128 x 128] + G3 [16 x 128] + db [2 x 128] + db3 [16]
    assert lib.orl_gt_raw_floats(C.byref(d)) == 128 * 16 + 128 * 128 + 16 * 128 + 2 * 128 + 16
    # image: resident part (fc1 [128 x 4], biases, head matrices) padded to 1 KB + 2 x 4 chunks of 26 KB
    chunk = (3 * 32 * (128 + 8) * 2 + 1023) // 1024 * 1024 // 4
    res = 128 * 4 + 2 * 128 + 16 + 128 * 20 + 16 * 132
    assert lib.orl_gt_image_floats(C.byref(d)) == (res + 255) // 256 * 256 + 8 * chunk
    for H, D, nl, ok in [(64, 64, 4, 1), (64, 4, 5, 0), (128, 4, 3, 1), (128, 4, 4, 1), (128, 4, 5, 0), (128, 32, 2, 1), (128, 64, 2, 0),
                         (96, 4, 2, 0), (256, 4, 2, 0), (64, 65, 2, 0)]:
        assert lib.orl_gt_supported(C.byref(desc(H, D, nl))) == ok, (H, D, nl)
    assert lib.orl_gt_supported(C.byref(desc(64, 4, 2, heads=(15, 1)))) == 1   # shared network: act + v_out
    assert lib.orl_gt_supported(C.byref(desc(64, 4, 2, heads=(16, 1)))) == 0   # 17 head outputs
    assert lib.orl_gt_raw_floats(C.byref(desc(256, 4, 2))) == -1
