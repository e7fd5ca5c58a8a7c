"""The oracle is pinned here: every entry point of oracle/oss_scan_oracle.c against the golden
vectors produced by the reference's own ``selective_scan_ref`` + autograd
(tests/golden/make_golden.py; reference Mamba/kernels/selective_scan/test_selective_scan.py:168-234).
Tolerances are the reference test's fp32 contract (test_selective_scan.py:398-401,490-502); the
oracle is in fact ~1e-6 relative."""
import pytest
import torch

from conftest import assert_close, golden_files, load_golden
from oracle import oss_oracle

RTOL, ATOL = 6e-4, 2e-3      # test_selective_scan.py:398
RTOLW, ATOLW = 1e-3, 1e-3    # :401


def _inputs(z):
    g = lambda k: z.get(k)
    return g("u"), g("delta"), g("A"), g("B"), g("C"), g("D"), g("delta_bias"), bool(z["delta_softplus"])


@pytest.mark.parametrize("name", golden_files("g1_scan_"))
@pytest.mark.parametrize("real", ["f32", "f64"])
def test_oracle_fwd_matches_reference(name, real):
    z = load_golden(name)
    u, dl, A, B, C, D, bias, sp = _inputs(z)
    out, x = oss_oracle.scan_fwd(u, dl, A, B, C, D, bias, sp, chunk=256, real=real)
    assert_close(out, z["out"], RTOL, ATOL, "out")
    assert_close(x[:, :, -1, 1::2], z["last_state"], RTOL, ATOL, "last_state")  # test_selective_scan.py:79
    # tight: fp32 restatement vs the fp32 reference agree to round-off
    assert_close(out, z["out"], 2e-5, 2e-5, "out (tight)")


@pytest.mark.parametrize("name", golden_files("g1_scan_"))
def test_oracle_bwd_matches_reference(name):
    z = load_golden(name)
    u, dl, A, B, C, D, bias, sp = _inputs(z)
    du, dd, dA, dB, dC, dD, db = oss_oracle.scan_bwd(u, dl, A, B, C, D, bias, z["dout"], None, sp)
    assert_close(du, z["du"], RTOL * 2, ATOL * 2, "du")              # :490
    assert_close(dd, z["ddelta"], RTOL * 5, ATOL * 10, "ddelta")     # :491
    assert_close(dA, z["dA"], RTOLW, ATOLW * 5, "dA")                # :492
    assert_close(dB, z["dB"], RTOL, ATOL, "dB")
    assert_close(dC, z["dC"], RTOL, ATOL, "dC")
    if D is not None:
        assert_close(dD, z["dD"], RTOLW, ATOLW, "dD")
    else:
        assert dD is None
    if bias is not None:
        assert_close(db, z["ddelta_bias"], RTOLW, ATOLW, "ddelta_bias")
    else:
        assert db is None


def test_oracle_chunk_states_are_consistent():
    """x[..., c, 1::2] is the state after chunk c whatever the chunk length; x[..., 0::2] is the
    running product of a (fwd_kernel.cuh:155-158)."""
    z = load_golden("g1_scan_twochunk2085_sp1_db1.npz")
    u, dl, A, B, C, D, bias, sp = _inputs(z)
    _, x256 = oss_oracle.scan_fwd(u, dl, A, B, C, D, bias, sp, chunk=256)
    _, x2048 = oss_oracle.scan_fwd(u, dl, A, B, C, D, bias, sp, chunk=2048)
    assert x256.shape[2] == 9 and x2048.shape[2] == 2
    assert torch.equal(x256[:, :, 7], x2048[:, :, 0])     # after step 2047
    assert torch.equal(x256[:, :, -1], x2048[:, :, -1])   # after the last step
    # restart from a saved state: scanning the tail from x256[3] reproduces the tail outputs
    t0 = 4 * 256
    h0 = x256[:, :, 3, 1::2]
    out_full, _ = oss_oracle.scan_fwd(u, dl, A, B, C, D, bias, sp, chunk=256)
    import torch.nn.functional as F
    dt = F.softplus(dl + bias[None, :, None])
    G = B.shape[1]
    rows = u.shape[1] // G
    y = torch.zeros_like(out_full[:, :, t0:])
    h = h0.clone()
    for t in range(t0, u.shape[2]):
        Bt = B[:, :, :, t].repeat_interleave(rows, dim=1)
        Ct = C[:, :, :, t].repeat_interleave(rows, dim=1)
        h = torch.exp(dt[:, :, t, None] * A[None]) * h + Bt * (dt[:, :, t] * u[:, :, t])[..., None]
        y[:, :, t - t0] = (h * Ct).sum(-1) + D[None] * u[:, :, t]
    assert_close(y, out_full[:, :, t0:], 1e-4, 1e-4, "restart from saved state")


def test_oracle_16bit_rounds_outputs_to_input_dtype():
    z = load_golden("g1_scan_bf16_s64_sp1_db1.npz")
    u, dl, A, B, C, D, bias, sp = _inputs(z)
    ub, db_, Bb, Cb = (t.to(torch.bfloat16) for t in (u, dl, B, C))
    assert torch.equal(ub.float(), u)  # fixtures hold exactly representable values
    out, _ = oss_oracle.scan_fwd(ub, db_, A, Bb, Cb, D, bias, sp)
    assert out.dtype == torch.bfloat16
    assert_close(out, z["out"], 3e-2, 5e-2, "bf16 out")  # test_selective_scan.py:399-400


@pytest.mark.parametrize("name", golden_files("g9_effn_"))
def test_effn_half_of_the_block_matches_reference(name, oracle_cpu_kernel):
    """G9 (round 6): ``x + ffn(norm2(x))`` of the reference's own MamberBlock (SRGAN / RealSR / mamber32 trees, WithBias and BiasFree
    LayerNorm) against the host-side mirror of this repo on the CPU twins, fp32: the fixture the one-launch GPU forward
    (csrc/oss_effn.hip) is held to in tests/test_effn_gpu.py is first shown to be what LayerNorm + FeedForward of this repo compute"""
    from vmambair_amd import oss_block
    z = load_golden(name)
    dim, ln = int(z["dim"]), str(z["ln"])
    norm = oss_block.LayerNorm(dim, ln)
    ff = oss_block.FeedForward(dim, 2.66, False)
    norm.load_state_dict({k[len("sd.norm2."):]: v.float() for k, v in z.items() if k.startswith("sd.norm2.")}, strict=True)
    ff.load_state_dict({k[len("sd.ffn."):]: v.float() for k, v in z.items() if k.startswith("sd.ffn.")}, strict=True)
    x = z["x"].float()
    with torch.no_grad():
        y = ff(x, pre_norm=norm)
    assert_close(y, z["y"], 1e-5, 1e-5 * float(z["y"].abs().max()), "x + ffn(norm2(x))")
