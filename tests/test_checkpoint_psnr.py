"""f4: checkpoint round trip in the reference's file format and the PSNR acceptance gate.

Fixtures (tests/golden/make_golden.py::make_g5, produced by RUNNING the reference):
  g5_ckpt_mambasisr6_d8.pth  written by the reference's ``BaseModel.save_network`` (base_model.py:213-244) from two
                             reference-built MambaSISR6 nets ('params' = seed 0, 'params_ema' = seed 1)
  g5_net_psnr.npz            forward of both through the reference arch + ``calculate_psnr(tensor2img(.), crop 4, Y)``
  g5_psnr.npz                ``calculate_psnr`` / ``tensor2img`` / ``to_y_channel`` on random images
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_close, load_golden
from vmambair_amd import checkpoint as ck
from vmambair_amd import metrics
from vmambair_amd.archs import MambaSISR6

CKPT = os.path.join(GOLDEN, "g5_ckpt_mambasisr6_d8.pth")


def small_net():
    return MambaSISR6(dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)


# ---- PSNR restatement vs the reference's function -------------------------------------------------------------------
def test_psnr_matches_reference_values():
    z = np.load(os.path.join(GOLDEN, "g5_psnr.npz"))
    n = 0
    for i in (0, 1):
        a, b = z[f"u8_{i}.a"], z[f"u8_{i}.b"]
        for crop in (0, 4):
            for yc in (0, 1):
                want = float(z[f"u8_{i}.psnr_c{crop}_y{yc}"])
                got = metrics.calculate_psnr(a, b, crop, "HWC", bool(yc))
                # Y mode: the reference's mean and log run in fp32 (to_y_channel returns fp32), ours accumulate in fp64
                tol = 2e-5 if yc else 1e-9
                assert abs(got - want) < tol, (i, crop, yc, got, want)
                chw = metrics.calculate_psnr(a.transpose(2, 0, 1), b.transpose(2, 0, 1), crop, "CHW", bool(yc))
                assert abs(chw - want) < tol
                n += 1
    ta, tb = torch.from_numpy(z["f32.a"]), torch.from_numpy(z["f32.b"])
    for crop in (0, 4):
        want = float(z[f"f32.psnr_c{crop}_y0"])
        assert abs(metrics.calculate_psnr(ta, tb, crop) - want) < 1e-9
    assert n == 8
    assert metrics.calculate_psnr(z["u8_0.a"], z["u8_0.a"], 4, "HWC", True) == float("inf")
    with pytest.raises(AssertionError):
        metrics.calculate_psnr(z["u8_0.a"], z["u8_1.a"], 0)
    with pytest.raises(ValueError):
        metrics.calculate_psnr(z["u8_0.a"], z["u8_0.b"], 0, "WHC")


def test_tensor2img_and_y_channel_match_reference():
    z = np.load(os.path.join(GOLDEN, "g5_psnr.npz"))
    got = metrics.tensor2img(torch.from_numpy(z["t2i.in"]))
    assert got.dtype == torch.uint8 and np.array_equal(got.numpy(), z["t2i.out"])   # bit-exact (clamp, half-even, BGR)
    y = metrics.to_y_channel(torch.from_numpy(z["y.in"]))
    assert y.dtype == torch.float32 and np.allclose(y.numpy(), z["y.out"], rtol=0, atol=2e-5)


# ---- checkpoint format ----------------------------------------------------------------------------------------------
def test_reference_checkpoint_loads_strict_and_reproduces_the_reference_forward(oracle_cpu_kernel):
    z = load_golden("g5_net_psnr.npz")
    for key, ykey, pkey in (("params", "y_params", "psnr_params"), ("params_ema", "y_ema", "psnr_ema")):
        net = small_net()
        res = ck.load_network(net, CKPT, strict=True, param_key=key)
        assert not res.missing_keys and not res.unexpected_keys
        with torch.no_grad():
            y = net(z["x"])
        assert_close(y, z[ykey], 1e-3, 1e-3, f"forward with '{key}'")
        got = metrics.validation_psnr(y, z["gt"], 4, True)
        assert abs(got - float(z[pkey])) < 1e-3, (got, float(z[pkey]))    # the PSNR gate: 1e-3 dB
    assert ck.load_for_inference(small_net(), CKPT) == "params_ema"       # RealESRGANer's preference


def test_save_network_writes_the_reference_format(tmp_path):
    ref = torch.load(CKPT, map_location="cpu", weights_only=True)
    a, b = small_net(), small_net()
    a.load_state_dict(ref["params"], strict=True)
    b.load_state_dict(ref["params_ema"], strict=True)
    path = ck.save_network([torch.nn.DataParallel(a), b], str(tmp_path / "models" / "net_g_5.pth"), ["params", "params_ema"])
    mine = torch.load(path, map_location="cpu", weights_only=True)
    assert list(mine) == ["params", "params_ema"]
    for key in mine:
        assert sorted(mine[key]) == sorted(ref[key]), "same names (load_state_dict ignores order), no 'module.' prefix"
        assert all(torch.equal(mine[key][k], ref[key][k]) for k in ref[key])
    assert ck.save_iteration(a, str(tmp_path), "net_g", -1).endswith("net_g_latest.pth")
    with pytest.raises(ValueError):
        ck.save_network([a, b], str(tmp_path / "x.pth"), "params")


def test_load_network_semantics(tmp_path):
    ref = torch.load(CKPT, map_location="cpu", weights_only=True)
    # (1) 'module.' prefixes are stripped; a file without 'params_ema' falls back to 'params' (base_model.py:295-306)
    p1 = str(tmp_path / "ddp.pth")
    torch.save({"params": {"module." + k: v for k, v in ref["params"].items()}}, p1)
    net = small_net()
    ck.load_network(net, p1, strict=True, param_key="params_ema")
    assert all(torch.equal(v, ref["params"][k]) for k, v in net.state_dict().items())
    # (2) param_key=None: the file IS the state dict
    p2 = str(tmp_path / "bare.pth")
    torch.save(dict(ref["params_ema"]), p2)
    net = small_net()
    ck.load_network(net, p2, strict=True, param_key=None)
    assert all(torch.equal(v, ref["params_ema"][k]) for k, v in net.state_dict().items())
    # (3) strict: a missing / unexpected / wrongly sized entry raises; non-strict sets size mismatches aside
    broken = dict(ref["params"])
    some = next(k for k in broken if k.endswith("in_conv.weight"))
    broken[some] = broken[some][:-1]
    broken["extra.weight"] = torch.zeros(1)
    missing = next(k for k in broken if k.endswith("A_logs"))
    del broken[missing]
    p3 = str(tmp_path / "broken.pth")
    torch.save({"params": broken}, p3)
    with pytest.raises(RuntimeError):
        ck.load_network(small_net(), p3, strict=True)
    net = small_net()
    before = net.state_dict()[some].clone()
    res = ck.load_network(net, p3, strict=False)
    assert missing in res.missing_keys and some in res.missing_keys
    assert "extra.weight" in res.unexpected_keys and some + ".ignore" in res.unexpected_keys
    assert torch.equal(net.state_dict()[some], before), "the wrongly sized tensor was not loaded"
    diff = ck.different_keys(small_net(), ck.read_state(p3), strict=False)
    assert diff["size_mismatch"] == [some] and missing in diff["missing_in_file"]


@pytest.mark.gpu
def test_reference_checkpoint_on_gpu_reproduces_fixture_psnr():
    """the acceptance gate on the HIP kernels: reference-written weights -> drop-in net on the MI355X -> validation PSNR
    (tensor2img, crop 4, Y channel) within 1e-3 dB of what the reference arch produced from the same file"""
    z = load_golden("g5_net_psnr.npz")
    for key, ykey, pkey in (("params", "y_params", "psnr_params"), ("params_ema", "y_ema", "psnr_ema")):
        net = small_net()
        ck.load_network(net, CKPT, strict=True, param_key=key)
        net.to("cuda:0")
        with torch.no_grad():
            y = net(z["x"].to("cuda:0"))
        assert_close(y, z[ykey], 1e-3, 1e-3, f"HIP forward with '{key}'")
        got = metrics.validation_psnr(y, z["gt"].to("cuda:0"), 4, True)
        assert abs(got - float(z[pkey])) < 1e-3, (got, float(z[pkey]))
    # a training-state round trip on the device: save from the GPU net, load into a fresh one, identical forward
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = ck.save_network(net, os.path.join(d, "net_g_latest.pth"))
        net2 = small_net()
        ck.load_network(net2, path, strict=True)
        net2.to("cuda:0")
        with torch.no_grad():
            assert torch.equal(net2(z["x"].to("cuda:0")), y)
