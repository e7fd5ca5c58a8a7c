"""GPU parity tests of the HIP selective scan, THROUGH THE C ABI (vmambair_amd.ops -> ctypes ->
libvmambair_oss.so), against the CPU oracle on the same seeded inputs and against the golden
vectors of the reference.

The grid is the reference's own (Mamba/kernels/selective_scan/test_selective_scan.py:365-502):
itype x seqlen x delta_bias x softplus x D x groups, inputs ``A = -0.5 rand``, ``delta = 0.5 rand``,
others ``randn``, seed 0 -- with ``dstate = 16`` (the only value the archs use) added to the
reference's ``dstate = 1``, odd / multi-chunk lengths, strided B/C views and ragged row tiles.
Tolerances are the reference's (:398-401,490-502) and are written next to each comparison.
"""
import itertools

import pytest
import torch

from conftest import assert_close, golden_files, load_golden
import vmambair_amd
from vmambair_amd import _capi
from oracle import oss_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

TOL = {  # (rtol, atol) per input type -- test_selective_scan.py:398-400
    torch.float32: (6e-4, 2e-3),
    torch.float16: (3e-3, 5e-3),
    torch.bfloat16: (3e-2, 5e-2),
}
RTOLW, ATOLW = 1e-3, 1e-3  # :401


def make_inputs(batch, dim, N, G, L, itype, has_D=True, has_bias=True, seed=0, delta_scale=0.5):
    g = torch.Generator().manual_seed(seed)
    A = -0.5 * torch.rand(dim, N, generator=g)
    B = torch.randn(batch, G, N, L, generator=g).to(itype)
    C = torch.randn(batch, G, N, L, generator=g).to(itype)
    D = torch.randn(dim, generator=g) if has_D else None
    bias = 0.5 * torch.rand(dim, generator=g) if has_bias else None
    u = torch.randn(batch, dim, L, generator=g).to(itype)
    delta = (delta_scale * torch.rand(batch, dim, L, generator=g)).to(itype)
    dout = torch.randn(batch, dim, L, generator=g).to(itype)
    return u, delta, A, B, C, D, bias, dout


def to_dev(ts):
    return [t.to(DEV) if t is not None else None for t in ts]



def _need_feature(bit, name):
    """the fused-delta / lane-state scan forms are part of every library since round 6 (csrc/oss_host.h): a library without them
    is a broken build, not a reason to skip"""
    assert _capi.has_feature(bit), f"libvmambair_oss.so lacks the scan form '{name}' (oss_scan_features() = {_capi.load().oss_scan_features()})"


def check_fwd_bwd(cpu_inputs, softplus, itype, fwd_variant=-1, bwd_variant=-1, tight=True):
    u, delta, A, B, C, D, bias, dout = cpu_inputs
    lib = _capi.load()
    lib.oss_scan_set_variant(fwd_variant, bwd_variant)
    try:
        du_, dl_, A_, B_, C_, D_, b_, g_ = to_dev(cpu_inputs)
        out, x = vmambair_amd.selective_scan_fwd(du_, dl_, A_, B_, C_, D_, b_, softplus, 1)
        grads = vmambair_amd.selective_scan_bwd(du_, dl_, A_, B_, C_, D_, b_, g_, x, softplus, 1)
        torch.cuda.synchronize()
    finally:
        lib.oss_scan_set_variant(-1, -1)
    chunk = vmambair_amd.scan_chunk()
    ref_out, ref_x = oss_oracle.scan_fwd(u, delta, A, B, C, D, bias, softplus, chunk=chunk)
    ref = oss_oracle.scan_bwd(u, delta, A, B, C, D, bias, dout, None, softplus)
    rtol, atol = TOL[itype]
    assert out.dtype == itype and x.dtype == torch.float32
    assert_close(out, ref_out, rtol, atol, "out")
    assert_close(x[..., 1::2], ref_x[..., 1::2], 6e-4, 2e-3, "x states")       # every saved state
    assert_close(x[..., 0::2], ref_x[..., 0::2], 1e-3, 1e-6, "x running product")
    names = ["du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias"]
    tols = [(rtol * 2, atol * 2), (rtol * 5, atol * 10), (RTOLW, ATOLW * 5), (rtol, atol), (rtol, atol),
            (RTOLW, ATOLW), (RTOLW, ATOLW)]  # :490-502
    for n, got, want, (rt, at) in zip(names, grads, ref, tols):
        if want is None:
            assert got is None, n
            continue
        if n in ("dA", "dD", "ddelta_bias"):
            # sums of batch*L terms of mixed sign: two fp32 summation orders differ by ~1e-6 of the
            # LARGEST entry (the reference's absolute 5e-3 assumes its dim=768/dstate=1 grid), and
            # 16-bit inputs add their rounding
            at = max(at, (2e-5 if itype == torch.float32 else 2e-3) * float(want.abs().max()))
        assert_close(got, want, rt, at, n)
    if tight and itype == torch.float32:
        # both are fp32 implementations of the same recurrence: they agree far inside the contract
        # (absolute part scaled by the largest magnitude: outputs are sums with cancellation)
        for nm, g_, r_ in (("out", out, ref_out), ("du", grads[0], ref[0]), ("dB", grads[3], ref[3]), ("dC", grads[4], ref[4])):
            assert_close(g_, r_, 1e-4, 3e-6 * float(r_.abs().max()) + 1e-6, nm + " (tight)")


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_files("g1_scan_"))
def test_golden_vectors(name):
    """HIP path against the vectors produced by the reference itself."""
    z = load_golden(name)
    itype = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}[str(z["itype"])]
    sp = bool(z["delta_softplus"])
    cast = lambda k: z[k].to(itype).to(DEV)
    opt = lambda k: z[k].to(DEV) if k in z else None
    u, dl, B, C, g = cast("u"), cast("delta"), cast("B"), cast("C"), cast("dout")
    A, D, bias = z["A"].to(DEV), opt("D"), opt("delta_bias")
    out, x = vmambair_amd.selective_scan_fwd(u, dl, A, B, C, D, bias, sp, 1)
    du, dd, dA, dB, dC, dD, db = vmambair_amd.selective_scan_bwd(u, dl, A, B, C, D, bias, g, x, sp, 1)
    rtol, atol = TOL[itype]
    assert_close(out, z["out"], rtol, atol, "out")
    assert_close(x[:, :, -1, 1::2], z["last_state"], rtol, atol, "last_state")  # test_selective_scan.py:79
    assert_close(du, z["du"], rtol * 2, atol * 2, "du")
    assert_close(dd, z["ddelta"], rtol * 5, atol * 10, "ddelta")
    wa = max(ATOLW * 5, (2e-5 if itype == torch.float32 else 2e-3) * float(z["dA"].abs().max()))
    assert_close(dA, z["dA"], RTOLW, wa, "dA")
    assert_close(dB, z["dB"], rtol, atol, "dB")
    assert_close(dC, z["dC"], rtol, atol, "dC")
    if D is not None:
        wd = ATOLW if itype == torch.float32 else max(ATOLW, 2e-3 * float(z["dD"].abs().max()))
        assert_close(dD, z["dD"], RTOLW, wd, "dD")
    if bias is not None:
        wb = ATOLW if itype == torch.float32 else max(ATOLW, 2e-3 * float(z["ddelta_bias"].abs().max()))
        assert_close(db, z["ddelta_bias"], RTOLW, wb, "ddelta_bias")


@pytest.mark.parametrize("itype", [torch.float32, torch.float16, torch.bfloat16], ids=["f32", "f16", "bf16"])
@pytest.mark.parametrize("seqlen", [64, 128, 256, 512, 1024, 2048, 4096])
@pytest.mark.parametrize("flags", list(itertools.product([False, True], repeat=3)),
                         ids=lambda f: "bias%d_sp%d_D%d" % tuple(int(v) for v in f))
def test_reference_grid(itype, seqlen, flags):
    """The reference's grid (test_selective_scan.py:365-369) at dstate 16, two groups."""
    has_bias, softplus, has_D = flags
    inputs = make_inputs(2, 16, 16, 2, seqlen, itype, has_D, has_bias)
    check_fwd_bwd(inputs, softplus, itype)


@pytest.mark.parametrize("N,G", [(1, 1), (1, 2), (16, 1), (16, 4), (24, 2), (40, 1)])
@pytest.mark.parametrize("seqlen", [100, 777])
def test_dstate_and_groups(N, G, seqlen):
    """dstate = 1 is the reference grid's own value; 24 / 40 cross the 16-state LDS tile."""
    check_fwd_bwd(make_inputs(2, 8 * G, N, G, seqlen, torch.float32), True, torch.float32)


@pytest.mark.parametrize("seqlen", [1, 3, 37, 255, 257, 511, 513, 1061, 2048 + 37])
@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_ragged_lengths(seqlen, itype):
    """odd lengths: unaligned rows (scalar path), partial chunks, several saved states"""
    check_fwd_bwd(make_inputs(1, 12, 16, 4, seqlen, itype), True, itype)


@pytest.mark.parametrize("rows_per_group", [1, 3, 4, 5, 9, 17, 48])
def test_ragged_row_tiles(rows_per_group):
    """rows per group that do not fill a workgroup's row tile (dc_inner = 4 channel scans, D = 48)"""
    G = 2
    check_fwd_bwd(make_inputs(2, rows_per_group * G, 16, G, 320, torch.float32), True, torch.float32)


@pytest.mark.parametrize("fv", [0, 1, 2, 3, 4, 6])
@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_every_forward_variant(fv, itype):
    check_fwd_bwd(make_inputs(2, 32, 16, 2, 1100, itype), True, itype, fwd_variant=fv)


@pytest.mark.parametrize("bv", [1, 10, 11, 12, 13])
@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_every_backward_variant(bv, itype):
    check_fwd_bwd(make_inputs(2, 32, 16, 2, 1100, itype), True, itype, bwd_variant=bv)


@pytest.mark.parametrize("bv", [10, 11, 12, 13])
@pytest.mark.parametrize("case", ["ragged_rows", "short", "odd_len_f16", "no_softplus_no_D", "wide_state"])
def test_round2_backward_kernel_cases(bv, case):
    """oss_scan_bwd_v2.h (lane-resident state scalars, register-prefetched tiles, one barrier per state): row tiles that do
    not divide the group, sequences below one chunk, lengths that are not a multiple of 4 (scalar partial stores), the
    optional inputs absent, and dstate > 64 (falls back to the round-1 kernel of the same row count)"""
    if case == "ragged_rows":
        check_fwd_bwd(make_inputs(2, 2 * 13, 16, 2, 700, torch.float32), True, torch.float32, bwd_variant=bv)
    elif case == "short":
        check_fwd_bwd(make_inputs(3, 16, 16, 4, 37, torch.float32), True, torch.float32, bwd_variant=bv)
    elif case == "odd_len_f16":
        check_fwd_bwd(make_inputs(1, 24, 16, 2, 1539, torch.float16), True, torch.float16, bwd_variant=bv)
    elif case == "no_softplus_no_D":
        check_fwd_bwd(make_inputs(2, 16, 16, 2, 530, torch.float32, has_D=False, has_bias=False), False, torch.float32, bwd_variant=bv)
    else:
        check_fwd_bwd(make_inputs(1, 8, 72, 2, 300, torch.float32), True, torch.float32, bwd_variant=bv)


@pytest.mark.parametrize("dstate", [1, 5, 16, 19])
@pytest.mark.parametrize("bv", [1, 10, 13])
def test_backward_two_states_at_a_time_with_odd_state_counts(dstate, bv):
    """the round-2 kernels walk the states of a staging batch in pairs (register double set): an odd count leaves a single
    state at the end of a batch, 19 = four batches of four and three left over; variant 1 = the round-1 kernel"""
    check_fwd_bwd(make_inputs(2, 16, dstate, 2, 700, torch.float32), True, torch.float32, bwd_variant=bv)


def test_channel_scan_shapes():
    """the channel-direction calls of the archs: u (B, 2*dc_inner, D), L = D in {48..384}"""
    for D in (48, 96, 192, 384):
        check_fwd_bwd(make_inputs(3, 8, 16, 2, D, torch.float32, seed=D), True, torch.float32)
    check_fwd_bwd(make_inputs(3, 2, 16, 2, 96, torch.float32, seed=7), True, torch.float32)  # RealSR: one row/dir


def test_strided_views_like_the_archs():
    """B, C arrive as split views of x_dbl (unit last stride, arbitrary outer strides) and u/delta as
    views with a padded row pitch (cus/selective_scan.cpp:83-88,197-199)."""
    torch.manual_seed(0)
    Bsz, K, D, L, R, N = 2, 4, 8, 200, 3, 16
    x_dbl = torch.randn(Bsz, K, R + 2 * N, L)
    big_u = torch.randn(Bsz, K * D, L + 24)
    big_d = 0.5 * torch.rand(Bsz, K * D, L + 8)
    A = -0.5 * torch.rand(K * D, N)
    Dv, bias = torch.randn(K * D), 0.5 * torch.rand(K * D)
    dout = torch.randn(Bsz, K * D, L)

    def views(xd, bu, bd):
        _, Bs, Cs = torch.split(xd, [R, N, N], dim=2)
        return bu[:, :, 8:8 + L], bd[:, :, 4:4 + L], Bs, Cs

    u, delta, Bs, Cs = views(x_dbl, big_u, big_d)
    ud, dd_, Bd, Cd = views(x_dbl.to(DEV), big_u.to(DEV), big_d.to(DEV))
    assert not Bd.is_contiguous() and Bd.stride(-1) == 1 and ud.stride(1) == L + 24
    Ad, Dd, bd, gd = (t.to(DEV) for t in (A, Dv, bias, dout))
    out, x = vmambair_amd.selective_scan_fwd(ud, dd_, Ad, Bd, Cd, Dd, bd, True, 1)
    grads = vmambair_amd.selective_scan_bwd(ud, dd_, Ad, Bd, Cd, Dd, bd, gd, x, True, 1)
    ref_out, _ = oss_oracle.scan_fwd(u, delta, A, Bs, Cs, Dv, bias, True, chunk=256)
    ref = oss_oracle.scan_bwd(u, delta, A, Bs, Cs, Dv, bias, dout, None, True)
    assert_close(out, ref_out, 6e-4, 2e-3, "out")
    for n, got, want in zip(["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"], grads, ref):
        assert_close(got, want, 3e-3, 2e-2, n)
    assert grads[3].is_contiguous() and grads[3].shape == Bs.shape


def test_softplus_threshold_branch():
    """delta + bias straddling 20 (selective_scan_fwd_kernel.cuh:115-118, bwd_kernel.cuh:228-241)"""
    g = torch.Generator().manual_seed(3)
    u, _, A, B, C, D, bias, dout = make_inputs(1, 8, 16, 2, 300, torch.float32)
    delta = 19.0 + 2.0 * torch.rand(1, 8, 300, generator=g)
    A = A * 0.05
    check_fwd_bwd((u, delta, A, B, C, D, bias, dout), True, torch.float32, tight=False)
    # and far below zero (softplus -> exp), the dt-bias initialisation regime of the archs
    delta = -8.0 + 4.0 * torch.rand(1, 8, 300, generator=g)
    check_fwd_bwd((u, delta, A * 20, B, C, D, bias, dout), True, torch.float32)


def test_empty_batch_and_argument_errors():
    u, delta, A, B, C, D, bias, dout = to_dev(make_inputs(2, 8, 16, 2, 64, torch.float32))
    out, x = vmambair_amd.selective_scan_fwd(u[:0], delta[:0], A, B[:0], C[:0], D, bias, True, 1)
    assert out.shape == (0, 8, 64) and x.shape == (0, 8, 1, 32)
    with pytest.raises(RuntimeError):  # dtype mismatch (selective_scan.cpp:169-172)
        vmambair_amd.selective_scan_fwd(u, delta.half(), A, B, C, D, bias, True, 1)
    with pytest.raises(RuntimeError):  # dim % n_groups (selective_scan.cpp:190)
        vmambair_amd.selective_scan_fwd(u[:, :7], delta[:, :7], A[:7], B, C, None, None, True, 1)
    with pytest.raises(RuntimeError):  # time axis must be contiguous (:183-184)
        vmambair_amd.selective_scan_fwd(u.transpose(1, 2).contiguous().transpose(1, 2), delta, A, B, C, D, bias, True, 1)
    with pytest.raises(RuntimeError):  # weights are fp32 (:168)
        vmambair_amd.selective_scan_fwd(u, delta, A.half(), B, C, D, bias, True, 1)
    with pytest.raises(RuntimeError):  # CPU tensor
        vmambair_amd.selective_scan_fwd(u.cpu(), delta, A, B, C, D, bias, True, 1)
    lib = _capi.load()
    assert lib.oss_scan_fwd(None, 0, None) == -1


def test_autograd_function_and_drop_in_module():
    """SelectiveScanFn on the GPU vs the oracle's autograd twin; 3-D B/C (one group) are lifted and
    squeezed back (MambaSISR6_arch.py:41-46,72-73); the drop-in module returns the same tensors."""
    import selective_scan_cuda_core as core
    torch.manual_seed(0)
    u, delta, A, B, C, D, bias, dout = make_inputs(2, 8, 16, 1, 300, torch.float32)
    B3, C3 = B[:, 0], C[:, 0]
    leaves_cpu = [t.clone().requires_grad_() for t in (u, delta, A, B3, C3, D, bias)]
    leaves_gpu = [t.clone().to(DEV).requires_grad_() for t in (u, delta, A, B3, C3, D, bias)]
    y_cpu = oss_oracle.selective_scan_fn(*leaves_cpu, True)
    y_gpu = vmambair_amd.selective_scan_fn(*leaves_gpu, True)
    assert_close(y_gpu, y_cpu, 6e-4, 2e-3, "y")
    y_cpu.backward(dout)
    y_gpu.backward(dout.to(DEV))
    for n, a, b in zip("u delta A B C D bias".split(), leaves_gpu, leaves_cpu):
        assert a.grad.shape == b.grad.shape
        assert_close(a.grad, b.grad, 3e-3, 2e-2, "grad " + n)
    out, x = core.fwd(*[t.detach() for t in leaves_gpu[:3]], leaves_gpu[3].detach().unsqueeze(1),
                      leaves_gpu[4].detach().unsqueeze(1), leaves_gpu[5].detach(), leaves_gpu[6].detach(), True, 1)
    assert torch.equal(out, y_gpu.detach())
    res = core.bwd(*[t.detach() for t in leaves_gpu[:3]], leaves_gpu[3].detach().unsqueeze(1),
                   leaves_gpu[4].detach().unsqueeze(1), None, None, dout.to(DEV), x, True, 1)
    assert len(res) == 7 and res[5] is None and res[6] is None


def test_reruns_are_stable():
    """No zero-filled outputs, no atomics anywhere: a second call on the same inputs gives the same
    results bit for bit."""
    ins = to_dev(make_inputs(2, 64, 16, 2, 1500, torch.float32))
    u, delta, A, B, C, D, bias, dout = ins
    o1, x1 = vmambair_amd.selective_scan_fwd(u, delta, A, B, C, D, bias, True, 1)
    o2, x2 = vmambair_amd.selective_scan_fwd(u, delta, A, B, C, D, bias, True, 1)
    assert torch.equal(o1, o2) and torch.equal(x1, x2)
    g1 = vmambair_amd.selective_scan_bwd(u, delta, A, B, C, D, bias, dout, x1, True, 1)
    g2 = vmambair_amd.selective_scan_bwd(u, delta, A, B, C, D, bias, dout, x1, True, 1)
    for i in range(7):
        assert torch.equal(g1[i], g2[i])


# ------------------------------------------------------------------------------------------------
# BASELINE.json's full sizes: size-independent properties + sampled rows against the oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_full_size_properties(itype):
    """largest call of config 2: u (8, 384, 4096), B/C (8, 4, 16, 4096)"""
    torch.manual_seed(0)
    Bsz, KD, N, G, L = 8, 384, 16, 4, 4096
    u = torch.randn(Bsz, KD, L, device=DEV).to(itype)
    delta = (0.5 * torch.rand(Bsz, KD, L, device=DEV)).to(itype)
    A = -0.5 * torch.rand(KD, N, device=DEV)
    Bm = torch.randn(Bsz, G, N, L, device=DEV).to(itype)
    Cm = torch.randn(Bsz, G, N, L, device=DEV).to(itype)
    D = torch.randn(KD, device=DEV)
    bias = 0.5 * torch.rand(KD, device=DEV)
    out, x = vmambair_amd.selective_scan_fwd(u, delta, A, Bm, Cm, D, bias, True, 1)
    rtol, atol = TOL[itype]
    # (1) causality: the first half of the outputs does not depend on the second half of the inputs,
    #     and the state saved at the cut equals the last state of the truncated call
    h = L // 2
    out_h, x_h = vmambair_amd.selective_scan_fwd(u[:, :, :h].contiguous(), delta[:, :, :h].contiguous(), A,
                                                  Bm[..., :h].contiguous(), Cm[..., :h].contiguous(), D, bias, True, 1)
    assert torch.equal(out[:, :, :h], out_h)
    assert torch.equal(x[:, :, : h // 256], x_h)
    # (2) batch-permutation equivariance (rows are independent)
    perm = torch.randperm(Bsz, device=DEV)
    out_p, _ = vmambair_amd.selective_scan_fwd(u[perm], delta[perm], A, Bm[perm], Cm[perm], D, bias, True, 1)
    assert torch.equal(out_p, out[perm])
    # (3) linearity in C and the skip term: scan(C -> 2C) - D u = 2 (scan(C) - D u)
    out2, _ = vmambair_amd.selective_scan_fwd(u, delta, A, Bm, (2 * Cm.float()).to(itype), None, bias, True, 1)
    out1, _ = vmambair_amd.selective_scan_fwd(u, delta, A, Bm, Cm, None, bias, True, 1)
    assert_close(out2, 2 * out1.float(), rtol, atol, "linearity in C")
    # (4) sampled rows against the oracle (one group's B/C, eight rows of it)
    b, g0, r0 = 5, 2, 40
    rows = slice(g0 * (KD // G) + r0, g0 * (KD // G) + r0 + 8)
    sub = [t.cpu() for t in (u[b:b + 1, rows], delta[b:b + 1, rows], A[rows], Bm[b:b + 1, g0:g0 + 1],
                             Cm[b:b + 1, g0:g0 + 1], D[rows], bias[rows])]
    ref_out, ref_x = oss_oracle.scan_fwd(*sub, True, chunk=256)
    assert_close(out[b:b + 1, rows], ref_out, rtol, atol, "sampled rows")
    assert_close(x[b:b + 1, rows][..., 1::2], ref_x[..., 1::2], 6e-4, 2e-3, "sampled states")
    # (5) backward: linear in dout, and anti-causal (gradients at t >= T do not see dout before T)
    dout = torch.randn(Bsz, KD, L, device=DEV).to(itype)
    g1 = vmambair_amd.selective_scan_bwd(u, delta, A, Bm, Cm, D, bias, dout, x, True, 1)
    g2 = vmambair_amd.selective_scan_bwd(u, delta, A, Bm, Cm, D, bias, (2 * dout.float()).to(itype), x, True, 1)
    for i, n in enumerate(["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"]):
        sc = float(g1[i].float().abs().max())
        assert_close(g2[i], 2 * g1[i].float(), 4 * rtol, 4 * atol + 1e-3 * sc, "bwd linearity " + n)
    dz = dout.clone()
    dz[:, :, :h] = 0
    g3 = vmambair_amd.selective_scan_bwd(u, delta, A, Bm, Cm, D, bias, dz, x, True, 1)
    assert torch.equal(g3[0][:, :, h:], g1[0][:, :, h:]) and torch.equal(g3[1][:, :, h:], g1[1][:, :, h:])
    # sampled rows of du / ddelta against the oracle
    subg = oss_oracle.scan_bwd(*sub, dout[b:b + 1, rows].cpu(), None, True)
    assert_close(g1[0][b:b + 1, rows], subg[0], rtol * 2, atol * 2, "sampled du")
    assert_close(g1[1][b:b + 1, rows], subg[1], rtol * 5, atol * 10, "sampled ddelta")


# ------------------------------------------------------------------------------------------------
# omni form: mirrored directions and shared u rows inside the kernels
# ------------------------------------------------------------------------------------------------
# forced backward variants: the shortest ragged and the longest length only (the grid is built, not pruned by skips)
_OMNI_CASES = [pytest.param(seqlen, bv, id=f"{name}-{seqlen}") for bv, name in ((-1, "auto"), (10, "v2_12"), (11, "v2_8"))
               for seqlen in ((64, 100, 513, 1024, 2085) if bv < 0 else (100, 2085))]


@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("rows", [8, 48])
@pytest.mark.parametrize("seqlen,bv", _OMNI_CASES)
def test_omni_scan_matches_materialised_directions(seqlen, itype, rows, bv):
    lib = _capi.load()
    lib.oss_scan_set_variant(-1, bv)
    try:
        _omni_scan_case(seqlen, itype, rows)
    finally:
        lib.oss_scan_set_variant(-1, -1)


def _omni_scan_case(seqlen, itype, rows):
    """omni_scan(x2, ...) == selective_scan(cat[x2, flip x2], flip of delta/B/C for k >= 2) with the
    outputs and gradients of the mirrored directions flipped back (reference data flow:
    MambaSISR6_arch.py:401-428)."""
    K, N, Bsz = 4, 16, 2
    g = torch.Generator().manual_seed(11)
    x2 = torch.randn(Bsz, 2 * rows, seqlen, generator=g).to(itype)
    delta = (0.5 * torch.rand(Bsz, K * rows, seqlen, generator=g)).to(itype)
    A = -0.5 * torch.rand(K * rows, N, generator=g)
    Bm = torch.randn(Bsz, K, N, seqlen, generator=g).to(itype)
    Cm = torch.randn(Bsz, K, N, seqlen, generator=g).to(itype)
    D = torch.randn(K * rows, generator=g)
    bias = 0.5 * torch.rand(K * rows, generator=g)
    dout = torch.randn(Bsz, K * rows, seqlen, generator=g).to(itype)

    def mirror(t, per):
        t = t.clone()
        t[:, 2 * per:] = t[:, 2 * per:].flip(-1)
        return t

    u4 = mirror(x2.repeat(1, 2, 1), rows)
    ref_out, ref_x = oss_oracle.scan_fwd(u4, mirror(delta, rows), A, mirror(Bm, 1), mirror(Cm, 1), D, bias, True, chunk=256)
    ref = oss_oracle.scan_bwd(u4, mirror(delta, rows), A, mirror(Bm, 1), mirror(Cm, 1), D, bias, mirror(dout, rows), None, True)
    dv = [t.to(DEV) for t in (x2, delta, A, Bm, Cm, D, bias)]
    out, x = vmambair_amd.selective_scan_fwd(*dv, True, 1, rev_group_start=2, u_row_mod=2 * rows)
    grads = vmambair_amd.selective_scan_bwd(*dv, dout.to(DEV), x, True, 1, rev_group_start=2, u_row_mod=2 * rows)
    rtol, atol = TOL[itype]
    # the same call with A handed over as A_log (A = -exp(A_log) in-kernel) and dout shared by k, k+2
    A_log = torch.log(-A).to(DEV)
    out_l, x_l = vmambair_amd.selective_scan_fwd(dv[0], dv[1], A_log, *dv[3:], True, 1, rev_group_start=2,
                                                  u_row_mod=2 * rows, a_log_form=True)
    # -exp(log(-A)) differs from A in the last fp32 bit: same results up to the I/O rounding
    assert_close(out_l, out, rtol, atol, "A_log form out")
    g_l = vmambair_amd.selective_scan_bwd(dv[0], dv[1], A_log, *dv[3:], dout.to(DEV), x_l, True, 1, rev_group_start=2,
                                          u_row_mod=2 * rows, a_log_form=True)
    assert_close(g_l[2], grads[2] * A.to(DEV), 1e-3, (1e-5 if itype == torch.float32 else 2e-3) * float((grads[2] * A.to(DEV)).abs().max()), "dA_log = dA * A")
    assert_close(g_l[0], grads[0], rtol * 2, atol * 2, "A_log form du")
    d2 = dout[:, :2 * rows].to(DEV)
    g_s = vmambair_amd.selective_scan_bwd(*dv, d2, x, True, 1, rev_group_start=2, u_row_mod=2 * rows, dout_row_mod=2 * rows)
    g_r = vmambair_amd.selective_scan_bwd(*dv, d2.repeat(1, 2, 1), x, True, 1, rev_group_start=2, u_row_mod=2 * rows)
    for a_, b_ in zip(g_s, g_r):
        assert torch.equal(a_, b_)
    rtol, atol = TOL[itype]
    assert_close(out, mirror(ref_out, rows), rtol, atol, "out")
    assert_close(x[..., 1::2], ref_x[..., 1::2], 6e-4, 2e-3, "states")
    assert_close(grads[0], mirror(ref[0], rows), rtol * 2, atol * 2, "du (per direction)")
    assert_close(grads[1], mirror(ref[1], rows), rtol * 5, atol * 10, "ddelta")
    wa = (2e-5 if itype == torch.float32 else 2e-3)
    assert_close(grads[2], ref[2], RTOLW, max(ATOLW * 5, wa * float(ref[2].abs().max())), "dA")
    assert_close(grads[3], mirror(ref[3], 1), rtol, atol, "dB")
    assert_close(grads[4], mirror(ref[4], 1), rtol, atol, "dC")
    assert_close(grads[5], ref[5], RTOLW, max(ATOLW, wa * float(ref[5].abs().max())), "dD")
    assert_close(grads[6], ref[6], RTOLW, max(ATOLW, wa * float(ref[6].abs().max())), "dbias")


def test_omni_block_path_equals_reference_data_flow_on_gpu():
    from vmambair_amd.oss_block import SS2D_1
    torch.manual_seed(0)
    m = SS2D_1(d_model=48, ssm_ratio=1, variant="srgan").to(DEV)
    x = torch.randn(2, 48, 16, 24, device=DEV)
    res = []
    for omni in (True, False):
        m.omni = omni
        m.zero_grad()
        xi = x.clone().requires_grad_()
        y = m.forward_core(xi)
        y.square().sum().backward()
        res.append((y.detach(), xi.grad, {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert_close(res[0][0], res[1][0], 1e-4, 1e-4, "y")
    assert_close(res[0][1], res[1][1], 1e-3, 1e-3, "dx")
    for k in res[1][2]:
        assert_close(res[0][2][k], res[1][2][k], 2e-3, 2e-4 * max(1.0, float(res[1][2][k].abs().max())), k)


# ------------------------------------------------------------------------------------------------
# f1: delta computed inside the scan from the rank-R factor
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("L,rows,R,bv", [(1024, 48, 3, -1), (1539, 13, 6, -1), (4096, 96, 6, -1), (700, 24, 8, 13), (2085, 12, 5, 11)])
def test_fused_delta_scan_matches_materialised_delta(itype, L, rows, R, bv):
    """``dt_weight`` form (include/vmambair_oss.h): delta = W_dt . z inside the kernels, gradient of z and of W_dt out of the
    backward -- against the same kernels fed the materialised delta (the archs' ``dts = einsum(dts, dt_projs_weight)``,
    MambaSISR6_arch.py:409-411) with the projection and its adjoint done by torch in fp32.  Omni form (4 groups, two mirrored,
    shared u rows) as SS2D_1 issues it."""
    _need_feature(_capi.FEATURE_FUSED_DT, "fused_dt")
    K, N, Bsz = 4, 16, 2
    Cc = R + 2 * N
    g = torch.Generator().manual_seed(5)
    x2 = torch.randn(Bsz, 2 * rows, L, generator=g).to(itype).to(DEV)
    xdbl = torch.randn(Bsz, K, Cc, L, generator=g).to(itype).to(DEV)
    xdbl[:, :, :R] *= 0.3
    W = (torch.randn(K * rows, R, generator=g) * 0.5).to(DEV)
    A_log = torch.log(0.5 + torch.rand(K * rows, N, generator=g)).to(DEV)
    D, bias = torch.randn(K * rows, generator=g).to(DEV), (0.5 * torch.rand(K * rows, generator=g)).to(DEV)
    dout = torch.randn(Bsz, 2 * rows, L, generator=g).to(itype).to(DEV)
    Bm, Cm = xdbl[:, :, R:R + N], xdbl[:, :, R + N:]
    z = xdbl[:, :, :R].float()
    delta = torch.einsum("kdr,bkrl->bkdl", W.view(K, rows, R), z).reshape(Bsz, K * rows, L)
    lib = _capi.load()
    lib.oss_scan_set_variant(-1, bv)
    try:
        kw = dict(rev_group_start=2, u_row_mod=2 * rows, a_log_form=True)
        out_f, x_f = vmambair_amd.selective_scan_fwd(x2, xdbl, A_log, Bm, Cm, D, bias, True, 1, dt_weight=W, **kw)
        out_u, x_u = vmambair_amd.selective_scan_fwd(x2, delta.to(itype), A_log, Bm, Cm, D, bias, True, 1, **kw)
        dxf = torch.full_like(xdbl, float("nan"))
        dxu = torch.zeros_like(xdbl)
        gf = vmambair_amd.selective_scan_bwd(x2, xdbl, A_log, Bm, Cm, D, bias, dout, x_f, True, 1, dout_row_mod=2 * rows,
                                             dbc_into=dxf, dt_weight=W, **kw)
        gu = vmambair_amd.selective_scan_bwd(x2, delta.to(itype), A_log, Bm, Cm, D, bias, dout, x_u, True, 1,
                                             dout_row_mod=2 * rows, dbc_into=dxu, **kw)
    finally:
        lib.oss_scan_set_variant(-1, -1)
    assert len(gf) == 8 and gf[1] is None and len(gu) == 7
    # the materialised delta is rounded to the I/O type on its way into the kernel, the fused one is not
    rtol, atol = TOL[itype]
    assert_close(out_f, out_u, rtol, atol, "out")
    assert_close(x_f[..., 1::2], x_u[..., 1::2], rtol, atol, "saved states")
    assert_close(gf[0], gu[0], rtol * 2, atol * 2, "du")
    ddelta = gu[1].float().view(Bsz, K, rows, L)
    dz_ref = torch.einsum("kdr,bkdl->bkrl", W.view(K, rows, R), ddelta)
    dW_ref = torch.einsum("bkdl,bkrl->kdr", ddelta, z).reshape(K * rows, R)
    sc = float(dz_ref.abs().max())
    assert torch.isfinite(dxf).all(), "every row of the x_dbl gradient is written"
    assert_close(dxf[:, :, :R], dz_ref, rtol * 5, atol * 10 + (2e-5 if itype == torch.float32 else 8e-3) * sc, "gradient of the dt factor")
    assert_close(dxf[:, :, R:], dxu[:, :, R:], rtol, atol, "dB / dC rows")
    wa = (2e-5 if itype == torch.float32 else 4e-3)
    assert_close(gf[7], dW_ref, RTOLW * 5, max(ATOLW * 5, wa * float(dW_ref.abs().max())), "gradient of dt_weight")
    for i, n in ((2, "dA_log"), (5, "dD"), (6, "dbias")):
        assert_close(gf[i], gu[i], RTOLW * 5, max(ATOLW * 5, wa * float(gu[i].abs().max())), n)
    # bit-stable reruns of the fused form as well
    dxf2 = torch.empty_like(xdbl)
    gf2 = vmambair_amd.selective_scan_bwd(x2, xdbl, A_log, Bm, Cm, D, bias, dout, x_f, True, 1, dout_row_mod=2 * rows,
                                          dbc_into=dxf2, dt_weight=W, **kw) if bv < 0 else None
    if gf2 is not None:
        assert torch.equal(dxf2, dxf) and torch.equal(gf2[7], gf[7]) and torch.equal(gf2[0], gf[0])


@pytest.mark.parametrize("dim,hw", [(48, (32, 32)), (96, (24, 40))])
def test_fused_delta_core_matches_materialised_core(dim, hw):
    """SS2DCoreFn with and without the fused delta (ops.core.FUSED_DT) under bf16: same output and gradients to bf16 rounding"""
    _need_feature(_capi.FEATURE_FUSED_DT, "fused_dt")
    from vmambair_amd import ops
    from vmambair_amd.oss_block import SS2D_1
    torch.manual_seed(0)
    m = SS2D_1(d_model=dim, ssm_ratio=1, variant="srgan").to(DEV)
    x = torch.randn(2, dim, *hw, device=DEV).to(torch.bfloat16)
    res = []
    # a fixed random cotangent: mean(LayerNorm(.)^2) is constant up to the affine part, so its gradient is rounding noise
    gw = torch.randn(2, dim, *hw, device=DEV)
    for fused in (True, False):
        keep, ops.core.FUSED_DT = ops.core.FUSED_DT, fused
        try:
            assert ops.fused_dt_supported(torch.bfloat16, 2, dim, m.dt_rank + 32, m.dt_rank, 16, hw[0] * hw[1]) == fused
            m.zero_grad()
            xi = x.clone().requires_grad_()
            y = m.forward_core(xi)
            (y.float() * gw).sum().backward()
        finally:
            ops.core.FUSED_DT = keep
        res.append((y.detach().float(), xi.grad.float(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))

    def rel(a, b):
        return float((a - b).norm() / b.norm().clamp_min(1e-20))
    assert rel(res[0][0], res[1][0]) < 1e-2 and rel(res[0][1], res[1][1]) < 2e-2
    for k in res[1][2]:
        assert rel(res[0][2][k], res[1][2][k]) < 3e-2, k


@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("L,rows,R,bv", [(1024, 48, 3, -1), (1539, 13, 6, -1), (2085, 12, 8, 11), (700, 24, 5, 13)])
def test_fused_delta_scan_against_oracle(itype, L, rows, R, bv):
    """f1 against the ORACLE, not against our own unfused kernels: the four materialised directions
    (reference data flow, MambaSISR6_arch.py:401-428) with delta = W_dt . z evaluated on the CPU WITHOUT rounding to the I/O
    type (what the fused kernels do in fp32), forward through oracle/oss_scan_oracle.c, backward through its float64 twin; the
    gradients of z and W_dt follow from the oracle's ddelta by the einsum's own adjoint."""
    _need_feature(_capi.FEATURE_FUSED_DT, "fused_dt")
    K, N, Bsz = 4, 16, 2
    Cc = R + 2 * N
    g = torch.Generator().manual_seed(17)
    x2 = torch.randn(Bsz, 2 * rows, L, generator=g).to(itype)
    xdbl = torch.randn(Bsz, K, Cc, L, generator=g)
    xdbl[:, :, :R] *= 0.3
    xdbl = xdbl.to(itype)
    W = torch.randn(K * rows, R, generator=g) * 0.5
    A_log = torch.log(0.5 + torch.rand(K * rows, N, generator=g))
    D, bias = torch.randn(K * rows, generator=g), 0.5 * torch.rand(K * rows, generator=g)
    dout = torch.randn(Bsz, 2 * rows, L, generator=g).to(itype)

    def mirror(t, per):
        t = t.clone()
        t[:, 2 * per:] = t[:, 2 * per:].flip(-1)
        return t
    # ---- oracle on the materialised directions
    A = -torch.exp(A_log)
    z4 = mirror(xdbl[:, :, :R].double(), 1)
    delta4 = torch.einsum("kdr,bkrl->bkdl", W.view(K, rows, R).double(), z4).reshape(Bsz, K * rows, L)
    u4, g4 = mirror(x2.repeat(1, 2, 1), rows), mirror(dout.repeat(1, 2, 1), rows)
    B4, C4 = mirror(xdbl[:, :, R:R + N], 1), mirror(xdbl[:, :, R + N:], 1)
    ref_out, ref_x = oss_oracle.scan_fwd(u4, delta4.float(), A, B4, C4, D, bias, True, chunk=256)
    r64 = oss_oracle.scan_bwd(u4, delta4, A, B4, C4, D, bias, g4, None, True, real="f64")
    dd4 = r64[1].view(Bsz, K, rows, L)
    dz_ref = mirror(torch.einsum("kdr,bkdl->bkrl", W.view(K, rows, R).double(), dd4), 1)
    dW_ref = torch.einsum("bkdl,bkrl->kdr", dd4, z4).reshape(K * rows, R)
    # ---- fused HIP kernels, omni form
    lib = _capi.load()
    lib.oss_scan_set_variant(-1, bv)
    try:
        dv = [t.to(DEV) for t in (x2, xdbl, A_log)]
        Bm, Cm = dv[1][:, :, R:R + N], dv[1][:, :, R + N:]
        kw = dict(rev_group_start=2, u_row_mod=2 * rows, a_log_form=True)
        out_f, x_f = vmambair_amd.selective_scan_fwd(dv[0], dv[1], dv[2], Bm, Cm, D.to(DEV), bias.to(DEV), True, 1,
                                                      dt_weight=W.to(DEV), **kw)
        dxf = torch.full_like(dv[1], float("nan"))
        gf = vmambair_amd.selective_scan_bwd(dv[0], dv[1], dv[2], Bm, Cm, D.to(DEV), bias.to(DEV), dout.to(DEV), x_f, True, 1,
                                             dout_row_mod=2 * rows, dbc_into=dxf, dt_weight=W.to(DEV), **kw)
    finally:
        lib.oss_scan_set_variant(-1, -1)
    rtol, atol = TOL[itype]
    lo = itype == torch.float32
    assert_close(out_f, mirror(ref_out, rows), rtol, atol, "out")
    assert_close(x_f[..., 1::2], ref_x[..., 1::2], 6e-4, 2e-3, "saved states")
    assert_close(gf[0], mirror(r64[0], rows), rtol * 2, atol * 2, "du")
    assert torch.isfinite(dxf).all(), "every row of the x_dbl gradient is written"
    sc = float(dz_ref.abs().max())
    assert_close(dxf[:, :, :R], dz_ref, rtol * 5, atol * 10 + (2e-5 if lo else 8e-3) * sc, "gradient of the dt factor z")
    assert_close(dxf[:, :, R:R + N], mirror(r64[3], 1), rtol, atol, "dB")
    assert_close(dxf[:, :, R + N:], mirror(r64[4], 1), rtol, atol, "dC")
    wa = 2e-5 if lo else 4e-3
    assert_close(gf[7], dW_ref, RTOLW * 5, max(ATOLW * 5, wa * float(dW_ref.abs().max())), "gradient of dt_weight")
    dAlog_ref = r64[2] * A.double()                         # A = -exp(A_log)  =>  d/dA_log = dA * A
    assert_close(gf[2], dAlog_ref, RTOLW * 5, max(ATOLW * 5, wa * float(dAlog_ref.abs().max())), "dA_log")
    assert_close(gf[5], r64[5], RTOLW * 5, max(ATOLW * 5, wa * float(r64[5].abs().max())), "dD")
    assert_close(gf[6], r64[6], RTOLW * 5, max(ATOLW * 5, wa * float(r64[6].abs().max())), "dbias")


@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_fused_delta_core_against_oracle_twin(itype, oracle_cpu_kernel):
    """SS2DCoreFn with delta evaluated inside the scans (ops.core.FUSED_DT) vs the literal reference data flow of
    SS2D_1.forward_core on the CPU oracle (oracle/cpu_twins.py: ss2d_core), same tensors"""
    _need_feature(_capi.FEATURE_FUSED_DT, "fused_dt")
    from vmambair_amd import ops
    from vmambair_amd.oss_block import SS2D_1
    torch.manual_seed(8)
    m = SS2D_1(d_model=48, ssm_ratio=1, variant="srgan")
    x = torch.randn(2, m.d_inner, 32, 24).to(itype).float()      # values representable in the I/O type
    gw = torch.randn(2, m.d_inner, 32, 24)
    xc = x.clone().requires_grad_()
    yc = ops.SS2DCoreFn.apply(xc, m.x_proj_weight, m.dt_projs_weight, m.A_logs, m.Ds, m.dt_projs_bias)   # CPU twin, fp32
    (yc * gw).sum().backward()
    want = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    m.zero_grad()
    md = m.to(DEV)
    keep, ops.core.FUSED_DT = ops.core.FUSED_DT, True
    try:
        if itype != torch.float32:   # the fused form is 16-bit only (oss_scan_fused_dt_ok); fp32 runs the plain core: same check
            assert ops.fused_dt_supported(itype, 2, m.d_inner, m.dt_rank + 32, m.dt_rank, 16, 32 * 24)
        xg = x.to(DEV).to(itype).requires_grad_()
        yg = ops.SS2DCoreFn.apply(xg, md.x_proj_weight, md.dt_projs_weight, md.A_logs, md.Ds, md.dt_projs_bias)
        (yg.float() * gw.to(DEV)).sum().backward()
    finally:
        ops.core.FUSED_DT = keep

    def rel(a, b):
        return float((a.float().cpu() - b).norm() / b.norm().clamp_min(1e-20))
    lim = 2e-4 if itype == torch.float32 else 2e-2      # bf16: x_dbl, B, C, out are rounded to bf16 between the kernels
    assert rel(yg, yc.detach()) < lim and rel(xg.grad, xc.grad) < 2 * lim
    for k, gref in want.items():
        assert rel(dict(md.named_parameters())[k].grad, gref) < 3 * lim, k


def test_large_dstate_falls_back_when_the_tiles_do_not_fit_lds():
    """dstate = 250 with the 12-row / 1024-step forward variant needs 167 KiB of LDS (> 160 KiB on gfx950): the launcher takes the
    small-shape variant instead of failing; the reference admits dstate <= 256 (selective_scan.cpp:191)"""
    check_fwd_bwd(make_inputs(1, 24, 250, 2, 1100, torch.float32), True, torch.float32, fwd_variant=6, bwd_variant=1)
    check_fwd_bwd(make_inputs(1, 24, 250, 2, 1100, torch.float32), True, torch.float32, fwd_variant=3, bwd_variant=10)


# ------------------------------------------------------------------------------------------------
# time-segmented launches (round 3): workgroup = (batch, group, row tile, SEGMENT); the reference walks a row's chunks
# sequentially in one block (cus/selective_scan_fwd_kernel.cuh:101-102,147-158, cus/selective_scan_bwd_kernel.cuh:120-125,184)
# ------------------------------------------------------------------------------------------------
def _with_segments(fs, bs, fn):
    lib = _capi.load()
    lib.oss_scan_set_segments(fs, bs)
    try:
        return fn(lib)
    finally:
        lib.oss_scan_set_segments(-1, -1)
        lib.oss_scan_set_variant(-1, -1)


# 16-bit types: 2, 3 and the maximum segment count; float also 5
_SEG_CASES = [pytest.param(it, sg, id=f"{name}-{sg}") for it, name in ((torch.float32, "f32"), (torch.bfloat16, "bf16"), (torch.float16, "f16"))
              for sg in ((2, 3, 5, 64) if it == torch.float32 else (2, 3, 64))]


@pytest.mark.parametrize("seqlen", [1024, 2048 + 37, 4096 + 3, 5000])
@pytest.mark.parametrize("itype,segs", _SEG_CASES)
def test_time_segmented_scans_match_oracle(itype, seqlen, segs):
    """every segment count (64 = one chunk per segment) at full, ragged and non-multiple-of-4 lengths, all seven gradients"""

    def run(lib):
        # forward variant 0 / backward variant 13: 512-step chunks both ways, so that every length here has >= 2 chunks
        check_fwd_bwd(make_inputs(2, 24, 16, 2, seqlen, itype), True, itype, fwd_variant=0, bwd_variant=13)
        want = min(segs, (seqlen + 511) // 512)
        assert lib.oss_scan_last_segments(0) == want and lib.oss_scan_last_segments(1) == want, "the segmented kernels did not run"
        check_fwd_bwd(make_inputs(2, 24, 16, 2, seqlen, itype), True, itype)   # whatever variants the heuristics pick
    _with_segments(segs, segs, run)


@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("segs,split", [(2, 1), (2, 2), (2, 3), (2, 4), (3, 2), (3, 3), (4, 2), (2, 0), (64, 4)])
@pytest.mark.parametrize("seqlen", [4096, 3700, 6 * 512 + 5, 5120, 6600])
def test_carry_pass_segmentation_finer_than_the_main_launch(itype, segs, split, seqlen):
    """(round 5) oss_scan_set_carry_split: the forward's local pass and the backward's reverse-carry pass run on `split` pieces per
    main segment (pieces of ceil(chunks / split) chunks, the last one shorter), the main launch folds one pair per piece.  Lengths:
    8 full chunks, a ragged last chunk, a last segment shorter than the others (7 chunks), 10 chunks (two segments of 5: pieces 3 + 2
    or 2 + 2 + 1), 13 chunks (segments of 7 and 6: the short one leaves its last piece EMPTY).  split 1 = rounds 2-4, 0 = heuristic;
    64 segments = one chunk per segment, nothing left to split.  All outputs and the seven gradients against the oracle."""
    def run(lib):
        lib.oss_scan_set_carry_split(split)
        try:
            check_fwd_bwd(make_inputs(2, 24, 16, 2, seqlen, itype, seed=5), True, itype, fwd_variant=0, bwd_variant=13)   # 512-step chunks both ways
            want = min(segs, (seqlen + 511) // 512)
            assert lib.oss_scan_last_segments(0) == want and lib.oss_scan_last_segments(1) == want
            check_fwd_bwd(make_inputs(1, 26, 16, 2, seqlen, itype, seed=6), True, itype, fwd_variant=6, bwd_variant=10)   # 1024 / 512 steps, ragged row tile
        finally:
            lib.oss_scan_set_carry_split(0)
    _with_segments(segs, segs, run)


def test_carry_split_changes_association_only():
    """the same call with split 1 and split 4: equal to fp32 round-off (the pairs are folded in the same order, grouped
    differently), bit-identical reruns at either setting"""
    lib = _capi.load()
    ins = to_dev(make_inputs(2, 48, 16, 4, 4096, torch.float32, seed=9))
    u, dl, A, B, C, D, b, g = ins

    def call(split):
        lib.oss_scan_set_carry_split(split)
        out, x = vmambair_amd.selective_scan_fwd(u, dl, A, B, C, D, b, True, 1)
        grads = vmambair_amd.selective_scan_bwd(u, dl, A, B, C, D, b, g, x, True, 1)
        torch.cuda.synchronize()
        return [out] + [t for t in grads if t is not None]
    lib.oss_scan_set_segments(2, 2)
    lib.oss_scan_set_variant(0, 13)
    try:
        a1, a1b, a4, a4b = call(1), call(1), call(4), call(4)
    finally:
        lib.oss_scan_set_segments(-1, -1)
        lib.oss_scan_set_variant(-1, -1)
        lib.oss_scan_set_carry_split(0)
    for x1, x1b, x4, x4b in zip(a1, a1b, a4, a4b):
        assert torch.equal(x1, x1b) and torch.equal(x4, x4b)
        assert_close(x4, x1, 2e-5, 2e-5 * max(1.0, float(x1.abs().max())), "split 4 vs split 1")


@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("L,rows,R,bv,segs", [(1024, 48, 3, -1, -1), (4096, 96, 6, -1, -1), (2052, 13, 8, 13, 3), (700, 24, 5, 1, -1), (516, 12, 1, 11, -1)])
def test_dt_factor_gradient_in_the_finishing_launch(itype, L, rows, R, bv, segs):
    """(round 6) ``finish_dt_weight`` (include/vmambair_oss.h: oss_scan_bwd_params): the backward's finishing launch also evaluates
    the adjoint of the archs' dt_proj einsum (MambaSISR6_arch.py:411), ddt[b, k, r, t] = sum_d W[k, d, r] ddelta[b, k, d, t], into the
    first R rows of the x_dbl gradient.  Checked against that sum evaluated in float64 from the SAME call's ddelta (the scan's own
    outputs are checked against the oracle elsewhere and must not change: bit-identical with and without the extra workgroups);
    round-1 and round-2 kernels, a time-segmented launch, ragged row tiles, every I/O type, rank 1 .. 8."""
    K, N, Bsz = 4, 16, 2
    Cc = R + 2 * N
    cpu = make_inputs(Bsz, K * rows, N, K, L, itype, seed=23)
    u, dl, A, B, C, D, b, g = to_dev(cpu)
    W = (torch.randn(K * rows, R, generator=torch.Generator().manual_seed(3)) * 0.5).to(DEV)
    out, x = vmambair_amd.selective_scan_fwd(u, dl, A, B, C, D, b, True, 1)
    tune = (bv if bv >= 0 else None, segs if segs > 0 else None, None)
    into_a = torch.zeros(Bsz, K, Cc, L, dtype=itype, device=DEV)
    into_b = torch.full((Bsz, K, Cc, L), 7.0, dtype=itype, device=DEV)
    plain = vmambair_amd.selective_scan_bwd(u, dl, A, B, C, D, b, g, x, True, 1, dbc_into=into_a, tune=tune)
    with_dt = vmambair_amd.selective_scan_bwd(u, dl, A, B, C, D, b, g, x, True, 1, dbc_into=into_b, tune=tune, finish_dt_weight=W)
    for n, p_, q_ in zip(["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"], plain, with_dt):
        assert torch.equal(p_, q_), n
    assert torch.equal(into_a[:, :, R:], into_b[:, :, R:]) and float(into_a[:, :, :R].abs().max()) == 0.0
    want = torch.einsum("bkdl,kdr->bkrl", with_dt[1].double().view(Bsz, K, rows, L), W.double().view(K, rows, R))
    got = into_b[:, :, :R].double()
    lo = itype == torch.float32
    scale = float(want.abs().max())
    assert_close(got, want, 1e-5 if lo else 8e-3, (1e-5 if lo else 4e-3) * scale, "dt rows of the x_dbl gradient")


@pytest.mark.parametrize("rows", [96, 48, 108])
def test_opt_in_bf16_row_tile_partials_change_only_the_row_tile_sums(rows):
    """(round 6) ``tune=(..., "bf16")`` (oss_scan_bwd_params.tune_partials = 2): at bf16 I/O the round-2 backward writes its dB / dC
    row-tile partials as bf16 (half the scratch traffic; DESIGN.md 4.2 says why it is NOT the default: one element of the
    reference's grid left the bf16 tolerance).  Everything that is not a sum over row tiles is bit-identical to the default;
    dB / dC differ by at most tiles x 2^-9 x the largest partial -- bounded here by the largest |dB| of the default run times the
    tile count.  The default is checked against the oracle everywhere else in this file."""
    G, N, L, Bsz = 2, 16, 1536, 2
    u, dl, A, B, C, D, b, g = to_dev(make_inputs(Bsz, G * rows, N, G, L, torch.bfloat16, seed=31))
    out, x = vmambair_amd.selective_scan_fwd(u, dl, A, B, C, D, b, True, 1)
    f32p = vmambair_amd.selective_scan_bwd(u, dl, A, B, C, D, b, g, x, True, 1, tune=(10, 1, None))
    bf16p = vmambair_amd.selective_scan_bwd(u, dl, A, B, C, D, b, g, x, True, 1, tune=(10, 1, None, "bf16"))
    names = ["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"]
    tiles = (rows + 11) // 12
    for n, a, f in zip(names, bf16p, f32p):
        if n in ("dB", "dC"):
            bound = tiles * 2.0 ** -8 * float(f.float().abs().max())
            d = float((a.float() - f.float()).abs().max())
            assert 0 < d <= bound, (n, d, bound)          # > 0: the bf16 form really ran
        else:
            assert torch.equal(a, f), n


@pytest.mark.parametrize("boundary", ["c++", "ctypes"])
def test_per_call_tuning_fields_select_the_launch_shape_without_global_state(boundary):
    """(round 6) ``oss_scan_fwd_params.tune_*`` / ``oss_scan_bwd_params.tune_*``: variant, time segments and the
    carry split of ONE call, with the process-global setters left at their defaults -- bit-identical to the same launch shape
    forced through the test-only setters, on both host boundaries; and a per-call field wins over a contradicting global."""
    from vmambair_amd import _host
    lib = _capi.load()
    u, dl, A, B, C, D, b, g = to_dev(make_inputs(2, 48, 16, 4, 4096, torch.float32, seed=21))

    def call(tune_f=None, tune_b=None):
        out, x = vmambair_amd.selective_scan_fwd(u, dl, A, B, C, D, b, True, 1, tune=tune_f)
        shape_f = (lib.oss_scan_last_variant(0), lib.oss_scan_last_segments(0))
        grads = vmambair_amd.selective_scan_bwd(u, dl, A, B, C, D, b, g, x, True, 1, tune=tune_b)
        shape_b = (lib.oss_scan_last_variant(1), lib.oss_scan_last_segments(1))
        torch.cuda.synchronize()
        return [out] + [t for t in grads if t is not None], shape_f, shape_b

    _host.use(boundary)
    try:
        per_call, sf, sb = call((0, 2, 4), (13, 2, 4))
        assert sf == (0, 2) and sb == (13, 2)
        heur, hf, hb = call()                                    # nothing sticks: the next call is back on the heuristics
        lib.oss_scan_set_segments(2, 2); lib.oss_scan_set_variant(0, 13); lib.oss_scan_set_carry_split(4)
        via_globals, gf, gb = call()
        assert gf == sf and gb == sb
        # a per-call field wins over the global: 1 = never segment, variant 4 / 1
        _, wf, wb = call((4, 1, None), (1, 1, None))
        assert wf == (4, 1) and wb == (1, 1)
    finally:
        lib.oss_scan_set_segments(-1, -1); lib.oss_scan_set_variant(-1, -1); lib.oss_scan_set_carry_split(0)
        _host.use(None)
    assert hf != sf or hb != sb, "the heuristic picked the forced shape: the test shows nothing"
    for a, c in zip(per_call, via_globals):
        assert torch.equal(a, c)
    for a, c in zip(per_call, heur):
        assert_close(a, c, 2e-5, 2e-5 * max(1.0, float(c.abs().max())), "forced launch shape vs heuristic")


@pytest.mark.parametrize("fv,bv", [(0, 10), (3, 11), (6, 12), (4, 13), (1, 13), (2, 10)])
def test_time_segments_on_every_kernel_variant(fv, bv):
    """forward variants differ in chunk length (256 / 512 / 1024) and rows per wave; backward: the four round-2 row-tile sizes.
    Ragged row tiles (13 rows per group), no softplus / D / bias on one leg."""
    def run(lib):
        lib.oss_scan_set_variant(fv, bv)
        # check_fwd_bwd resets the variant override at its end, so force it inside via its arguments
        check_fwd_bwd(make_inputs(2, 26, 16, 2, 3000, torch.float32), True, torch.float32, fwd_variant=fv, bwd_variant=bv)
        assert lib.oss_scan_last_segments(0) > 1 and lib.oss_scan_last_segments(1) > 1
        check_fwd_bwd(make_inputs(1, 16, 5, 4, 2100, torch.float32, has_D=False, has_bias=False), False, torch.float32,
                      fwd_variant=fv, bwd_variant=bv)
    _with_segments(3, 4, run)


@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("segs", [2, 7])
def test_time_segments_in_the_omni_form(itype, segs):
    """time-mirrored groups (rev_group_start), shared u / dout rows, A_log form: the calls the blocks make"""
    _with_segments(segs, segs, lambda lib: (_omni_scan_case(2085, itype, 8), _omni_scan_case(3000, itype, 48)))


def test_time_segments_large_dstate_and_fallbacks():
    """dstate 40 (three 16-state tiles), dstate 72 (> 64: the backward takes the unsegmented round-1 kernel, the forward still
    segments), a forward workspace that is too small (falls back to the unsegmented launch, same results)"""
    def run(lib):
        check_fwd_bwd(make_inputs(1, 8, 40, 2, 2500, torch.float32), True, torch.float32, bwd_variant=13)
        assert lib.oss_scan_last_segments(0) > 1 and lib.oss_scan_last_segments(1) > 1
        check_fwd_bwd(make_inputs(1, 8, 72, 2, 1500, torch.float32), True, torch.float32, bwd_variant=10)
        assert lib.oss_scan_last_segments(1) == 1
    _with_segments(4, 4, run)


def test_segmented_and_unsegmented_launches_agree_and_are_stable():
    """same inputs with 1 and with 4 segments: equal to fp32 round-off (a segment's entering state is the fold of
    segment-local states, a different association of the same products); reruns of the segmented launch are bit-identical"""
    ins = to_dev(make_inputs(2, 48, 16, 4, 4096, torch.float32))
    u, delta, A, B, C, D, bias, dout = ins

    def both(lib):
        o, x = vmambair_amd.selective_scan_fwd(u, delta, A, B, C, D, bias, True, 1)
        g = vmambair_amd.selective_scan_bwd(u, delta, A, B, C, D, bias, dout, x, True, 1)
        return o, x, g, lib.oss_scan_last_segments(0), lib.oss_scan_last_segments(1)
    o1, x1, g1, sf1, sb1 = _with_segments(1, 1, both)
    o4, x4, g4, sf4, sb4 = _with_segments(4, 4, both)
    o4b, x4b, g4b, _, _ = _with_segments(4, 4, both)
    assert (sf1, sb1) == (1, 1) and sf4 == 4 and sb4 == 4
    assert torch.equal(o4, o4b) and torch.equal(x4, x4b) and all(torch.equal(a, b) for a, b in zip(g4, g4b))
    assert_close(o4, o1, 1e-5, 1e-5 * float(o1.abs().max()), "out: 4 segments vs 1")
    assert_close(x4[..., 1::2], x1[..., 1::2], 1e-5, 1e-5 * float(x1[..., 1::2].abs().max()), "saved states")
    for n, a, b in zip(["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"], g4, g1):
        assert_close(a, b, 1e-4, 2e-5 * float(b.abs().max()), n + ": 4 segments vs 1")


def test_segment_heuristic_fills_idle_cus_only():
    """the default picks segments for under-filled launches (RealSR tiles at batch 1, Deraining level 0 at batch 4) and leaves
    the headline launches (one workgroup per CU already) alone"""
    lib = _capi.load()

    def segs(Bsz, KD, G, L, itype=torch.bfloat16):
        ins = to_dev(make_inputs(Bsz, KD, 16, G, L, itype))
        u, delta, A, B, C, D, bias, dout = ins
        o, x = vmambair_amd.selective_scan_fwd(u, delta, A, B, C, D, bias, True, 1)
        vmambair_amd.selective_scan_bwd(u, delta, A, B, C, D, bias, dout, x, True, 1)
        torch.cuda.synchronize()
        return lib.oss_scan_last_segments(0), lib.oss_scan_last_segments(1)
    assert segs(8, 384, 4, 4096) == (1, 1)
    f, b = segs(1, 384, 4, 128 * 128)
    assert f > 1 and b > 1
    f, b = segs(4, 192, 4, 128 * 128)
    assert f > 1 and b > 1


@pytest.mark.parametrize("itype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_segmented_scan_at_inference_tile_length(itype):
    """RealSR tile: batch 1, u (1, 384, 160*160) -- sampled rows of out / du / ddelta against the oracle, dB / dC of one
    group against the oracle on that group's rows"""
    torch.manual_seed(3)
    Bsz, KD, N, G, L = 1, 384, 16, 4, 160 * 160
    u = torch.randn(Bsz, KD, L, device=DEV).to(itype)
    delta = (0.5 * torch.rand(Bsz, KD, L, device=DEV)).to(itype)
    A = -0.5 * torch.rand(KD, N, device=DEV)
    Bm = torch.randn(Bsz, G, N, L, device=DEV).to(itype)
    Cm = torch.randn(Bsz, G, N, L, device=DEV).to(itype)
    D = torch.randn(KD, device=DEV)
    bias = 0.5 * torch.rand(KD, device=DEV)
    dout = torch.randn(Bsz, KD, L, device=DEV).to(itype)
    lib = _capi.load()
    out, x = vmambair_amd.selective_scan_fwd(u, delta, A, Bm, Cm, D, bias, True, 1)
    assert lib.oss_scan_last_segments(0) > 1
    g = vmambair_amd.selective_scan_bwd(u, delta, A, Bm, Cm, D, bias, dout, x, True, 1)
    assert lib.oss_scan_last_segments(1) > 1
    rtol, atol = TOL[itype]
    g0 = 1
    rows = slice(g0 * (KD // G), (g0 + 1) * (KD // G))      # every row of group 1: dB / dC need all of them
    sub = [t.cpu() for t in (u[:, rows], delta[:, rows], A[rows], Bm[:, g0:g0 + 1], Cm[:, g0:g0 + 1], D[rows], bias[rows])]
    ref_out, ref_x = oss_oracle.scan_fwd(*sub, True, chunk=256)
    ref = oss_oracle.scan_bwd(*sub, dout[:, rows].cpu(), None, True)
    assert_close(out[:, rows], ref_out, rtol, atol, "out")
    assert_close(x[:, rows][..., 1::2], ref_x[..., 1::2], 6e-4, 2e-3, "states")
    assert_close(g[0][:, rows], ref[0], rtol * 2, atol * 2, "du")
    assert_close(g[1][:, rows], ref[1], rtol * 5, atol * 10, "ddelta")
    assert_close(g[3][:, g0:g0 + 1], ref[3], rtol, atol * 4, "dB (96-row sums)")
    assert_close(g[4][:, g0:g0 + 1], ref[4], rtol, atol * 4, "dC (96-row sums)")
    assert_close(g[2][rows], ref[2], RTOLW, max(ATOLW * 5, 2e-3 * float(ref[2].abs().max())), "dA")


# ------------------------------------------------------------------------------------------------
# the two host boundaries over the same C ABI: compiled TORCH_LIBRARY layer (default) and ctypes
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_compiled_and_ctypes_boundaries_agree(itype):
    """csrc_host/oss_torch_host.cpp (what cus/selective_scan.cpp:157-349 is to the reference) and the ctypes marshalling in
    ops/scan.py fill the same parameter struct: bit-identical tensors, same None pattern, same errors; strided B / C views,
    omni arguments and the in-place dB / dC rows included"""
    from vmambair_amd import _host
    assert _host.mode() == "c++", "the compiled boundary must be the default on a GPU box (build() makes it)"
    torch.manual_seed(0)
    Bsz, K, rows, L, R, N = 2, 4, 12, 1300, 3, 16
    xdbl = torch.randn(Bsz, K, R + 2 * N, L, device=DEV).to(itype)
    x2 = torch.randn(Bsz, 2 * rows, L, device=DEV).to(itype)
    delta = (0.5 * torch.rand(Bsz, K * rows, L, device=DEV)).to(itype)
    A_log = torch.log(0.5 + torch.rand(K * rows, N, device=DEV))
    D, bias = torch.randn(K * rows, device=DEV), 0.5 * torch.rand(K * rows, device=DEV)
    dout = torch.randn(Bsz, 2 * rows, L, device=DEV).to(itype)
    Bm, Cm = xdbl[:, :, R:R + N], xdbl[:, :, R + N:]
    kw = dict(rev_group_start=2, u_row_mod=2 * rows, a_log_form=True)
    res = {}
    for mode in ("c++", "ctypes"):
        _host.use(mode)
        try:
            out, x = vmambair_amd.selective_scan_fwd(x2, delta, A_log, Bm, Cm, D, bias, True, 1, **kw)
            into = torch.zeros_like(xdbl)
            g = vmambair_amd.selective_scan_bwd(x2, delta, A_log, Bm, Cm, D, bias, dout, x, True, 1, dout_row_mod=2 * rows,
                                                dbc_into=into, **kw)
            plain = vmambair_amd.selective_scan_bwd(x2.repeat(1, 2, 1), delta, -torch.exp(A_log), Bm.contiguous(), Cm.contiguous(),
                                                    None, None, dout.repeat(1, 2, 1), None if L <= 256 else x, False, 1)
            with pytest.raises(RuntimeError, match="delta, B, C must have u's dtype"):
                vmambair_amd.selective_scan_fwd(x2, delta.float() if itype != torch.float32 else delta.half(), A_log, Bm, Cm, D, bias, True, 1, **kw)
            with pytest.raises(RuntimeError, match="dout must have u's shape"):
                vmambair_amd.selective_scan_bwd(x2, delta, A_log, Bm, Cm, D, bias, dout[:, :5], x, True, 1, dout_row_mod=2 * rows, **kw)
            e_out, e_x = vmambair_amd.selective_scan_fwd(x2[:0], delta[:0], A_log, Bm[:0], Cm[:0], D, bias, True, 1, **kw)
            assert e_out.shape == (0, K * rows, L) and e_x.shape[0] == 0
        finally:
            _host.use(None)
        res[mode] = (out, x, g, into, plain)
    a, b = res["c++"], res["ctypes"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
    for ga, gb in ((a[2], b[2]), (a[4], b[4])):
        assert len(ga) == len(gb) == 7
        for t1, t2 in zip(ga, gb):
            assert (t1 is None) == (t2 is None)
            if t1 is not None:
                assert t1.dtype == t2.dtype and t1.shape == t2.shape and torch.equal(t1, t2)
    assert a[4][5] is None and a[4][6] is None, "absent D / delta_bias come back as None through both boundaries"
    # dB / dC of the in-place form are views of dbc_into's last rows in both
    assert a[2][3].data_ptr() == a[3][:, :, R:R + N].data_ptr() and a[2][4].data_ptr() == a[3][:, :, R + N:].data_ptr()


# ------------------------------------------------------------------------------------------------
# lane states (round 3): the forward pass saves the state entering every 8-step block, the backward loads it instead of
# re-running the forward recurrence (the reference recomputes from the chunk state, cus/selective_scan_bwd_kernel.cuh:184-202)
# ------------------------------------------------------------------------------------------------
def _fwd_bwd_with_lane_states(cpu_inputs, softplus, fv=-1, bv=-1, segs=(-1, -1), **kw):
    lib = _capi.load()
    lib.oss_scan_set_variant(fv, bv)
    lib.oss_scan_set_segments(*segs)
    try:
        dv = to_dev(cpu_inputs)
        u, delta, A, B, C, D, bias, dout = dv
        fkw = {k: v for k, v in kw.items() if k != "dout_row_mod"}
        if "dout_row_mod" in kw:
            dout = dout[:, :kw["dout_row_mod"]].contiguous()
        out, x, hs = vmambair_amd.selective_scan_fwd(u, delta, A, B, C, D, bias, softplus, 1, want_hs=True, **fkw)
        g_hs = vmambair_amd.selective_scan_bwd(u, delta, A, B, C, D, bias, dout, x, softplus, 1, hs=hs, **kw)
        used = lib.oss_scan_last_lane_states()
        g_re = vmambair_amd.selective_scan_bwd(u, delta, A, B, C, D, bias, dout, x, softplus, 1, **kw)
        assert lib.oss_scan_last_lane_states() == 0
        torch.cuda.synchronize()
    finally:
        lib.oss_scan_set_variant(-1, -1)
        lib.oss_scan_set_segments(-1, -1)
    return out, x, hs, g_hs, g_re, used


# heuristic variants: every type and length; forced variants: f32 / bf16 at the ragged lengths
_HS_CASES = [pytest.param(it, L, fv, bv, id=f"{name}-{L}-f{fv}b{bv}")
             for fv, bv in ((-1, -1), (0, 10), (3, 11), (6, 12), (1, 10), (2, 11), (4, 13))
             for it, name in ((torch.float32, "f32"), (torch.bfloat16, "bf16"), (torch.float16, "f16"))
             for L in (300, 1024, 2085, 4096 + 3)
             if (fv, bv) == (-1, -1) or (it != torch.float16 and L != 1024)]


@pytest.mark.parametrize("itype,seqlen,fv,bv", _HS_CASES)
def test_backward_from_saved_lane_states_matches_oracle(itype, seqlen, fv, bv):
    """every forward variant writes the lane states (chunk lengths 256 / 512 / 1024, 4 / 8 / 16 steps per lane, several rows per
    wave), every round-2 backward variant reads them: all seven gradients against the oracle at the reference's tolerances, and
    against the recomputing kernels to fp32 round-off"""
    _need_feature(_capi.FEATURE_LANE_STATES, "lane_states")
    cpu = make_inputs(2, 26, 16, 2, seqlen, itype)
    out, x, hs, g_hs, g_re, used = _fwd_bwd_with_lane_states(cpu, True, fv, bv)
    u, delta, A, B, C, D, bias, dout = cpu
    assert used == 1, "the lane-state kernels did not run"
    assert hs.numel() == 2 * 26 * 16 * (((seqlen + 7) // 8 + 63) // 64 * 64)
    # the saved entries themselves: h entering block k = the oracle's state after step 8k - 1 (chunk = 8 -> one state per block)
    _, ref_x8 = oss_oracle.scan_fwd(u, delta, A, B, C, D, bias, True, chunk=8)
    L8 = (seqlen + 7) // 8
    stride = hs.numel() // (2 * 26 * 16)
    got = hs.view(2, 26, 16, stride)[..., 1:L8].cpu()                       # entries 1 .. L8-1 = states after blocks 0 .. L8-2
    want = ref_x8[:, :, :L8 - 1, 1::2].permute(0, 1, 3, 2)
    assert_close(got, want, 6e-4, 2e-3, "lane states")
    assert float(hs.view(2, 26, 16, stride)[..., 0].abs().max()) == 0.0, "the state entering step 0 is 0"
    ref = oss_oracle.scan_bwd(u, delta, A, B, C, D, bias, dout, None, True)
    rtol, atol = TOL[itype]
    tols = [(rtol * 2, atol * 2), (rtol * 5, atol * 10), (RTOLW, ATOLW * 5), (rtol, atol), (rtol, atol), (RTOLW, ATOLW), (RTOLW, ATOLW)]
    for n, got_, re_, want_, (rt, at) in zip(["du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias"], g_hs, g_re, ref, tols):
        if n in ("dA", "dD", "ddelta_bias"):
            at = max(at, (2e-5 if itype == torch.float32 else 2e-3) * float(want_.abs().max()))
        assert_close(got_, want_, rt, at, n)
        sc = float(re_.float().abs().max())
        assert_close(got_, re_, 1e-4 if itype == torch.float32 else 2e-2, (2e-5 if itype == torch.float32 else 8e-3) * sc + 1e-7,
                     n + ": saved vs recomputed forward states")


@pytest.mark.parametrize("segs", [(1, 1), (3, 4), (64, 64)])
def test_lane_states_with_time_segments_and_the_omni_form(segs):
    """segmented forward (real pass) writes them, segmented backward reads them; time-mirrored groups index them by SCAN position"""
    _need_feature(_capi.FEATURE_LANE_STATES, "lane_states")
    K, N, Bsz, rows, L = 4, 16, 2, 12, 3000
    g = torch.Generator().manual_seed(21)
    x2 = torch.randn(Bsz, 2 * rows, L, generator=g)
    delta = 0.5 * torch.rand(Bsz, K * rows, L, generator=g)
    A = -0.5 * torch.rand(K * rows, N, generator=g)
    Bm, Cm = torch.randn(Bsz, K, N, L, generator=g), torch.randn(Bsz, K, N, L, generator=g)
    D, bias = torch.randn(K * rows, generator=g), 0.5 * torch.rand(K * rows, generator=g)
    dout = torch.randn(Bsz, 2 * rows, L, generator=g)

    def mirror(t, per):
        t = t.clone()
        t[:, 2 * per:] = t[:, 2 * per:].flip(-1)
        return t
    out, x, hs, g_hs, g_re, used = _fwd_bwd_with_lane_states((x2, delta, A, Bm, Cm, D, bias, dout), True, segs=segs,
                                                             rev_group_start=2, u_row_mod=2 * rows, dout_row_mod=2 * rows)
    assert _capi.load().oss_scan_last_segments(1) >= 1
    assert used == 1
    u4, g4 = mirror(x2.repeat(1, 2, 1), rows), mirror(dout.repeat(1, 2, 1), rows)
    ref_out, _ = oss_oracle.scan_fwd(u4, mirror(delta, rows), A, mirror(Bm, 1), mirror(Cm, 1), D, bias, True, chunk=256)
    ref = oss_oracle.scan_bwd(u4, mirror(delta, rows), A, mirror(Bm, 1), mirror(Cm, 1), D, bias, g4, None, True)
    assert_close(out, mirror(ref_out, rows), 6e-4, 2e-3, "out")
    assert_close(g_hs[0], mirror(ref[0], rows), 1.2e-3, 4e-3, "du")
    assert_close(g_hs[1], mirror(ref[1], rows), 3e-3, 2e-2, "ddelta")
    assert_close(g_hs[3], mirror(ref[3], 1), 6e-4, 2e-3, "dB")
    assert_close(g_hs[4], mirror(ref[4], 1), 6e-4, 2e-3, "dC")
    assert_close(g_hs[2], ref[2], RTOLW, max(ATOLW * 5, 2e-5 * float(ref[2].abs().max())), "dA")


def test_lane_states_fall_back_where_they_do_not_apply():
    """dstate > 64 (round-1 backward kernel) and the fused-delta form ignore the buffer; a wrong-sized buffer is an error"""
    _need_feature(_capi.FEATURE_LANE_STATES, "lane_states")
    cpu = make_inputs(1, 8, 72, 2, 700, torch.float32)
    out, x, hs, g_hs, g_re, used = _fwd_bwd_with_lane_states(cpu, True)
    assert used == 0
    for a, b in zip(g_hs, g_re):
        assert torch.equal(a, b)
    u, delta, A, B, C, D, bias, dout = to_dev(make_inputs(1, 8, 16, 2, 700, torch.float32))
    o, x, hs = vmambair_amd.selective_scan_fwd(u, delta, A, B, C, D, bias, True, 1, want_hs=True)
    with pytest.raises(RuntimeError, match="lane-state tensor"):
        vmambair_amd.selective_scan_bwd(u, delta, A, B, C, D, bias, dout, x, True, 1, hs=hs[:-64])


def test_shipped_library_has_both_runtime_selected_scan_forms():
    """(round 6) delta-inside-the-scan and the lane states are compiled into the shipped library and chosen per
    call (dt_weight / hs): the driver's run exercises them instead of skipping 110 tests"""
    lib = _capi.load()
    assert lib.oss_scan_features() == (_capi.FEATURE_FUSED_DT | _capi.FEATURE_LANE_STATES)
    assert lib.oss_scan_lane_state_floats(1, 8, 700, 16) > 0
    assert lib.oss_scan_fused_dt_ok(_capi.OSS_BF16, 8, 96, 38, 6, 16, 4096) == 1
    assert lib.oss_scan_fused_dt_ok(_capi.OSS_BF16, 8, 384, 56, 24, 16, 4096) == 0      # dt_rank > 8 stays on the materialised delta
