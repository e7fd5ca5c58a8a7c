"""Second pin of the oracle, independent of the committed golden vectors: on random small shapes (ragged lengths, one
or several chunks of saved states, 1..16 states, with / without D, bias, softplus) the C restatement
(oracle/oss_scan_oracle.c) must agree with a float64 step-by-step evaluation of SURVEY.md Appendix A written here in
plain torch, and its backward with torch.autograd through that evaluation.  (The golden-vector tests pin the oracle to
the reference itself; this one pins every shape the GPU parity tests feed it.)"""
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

from conftest import assert_close
from oracle import oss_oracle


def scan_f64(u, delta, A, B, C, D, bias, softplus):
    """y, last state -- Appendix A, forward: per row r = (b, d), group g = d // (dim / G), sequentially over t"""
    u, delta, A, B, C = (t.double() for t in (u, delta, A, B, C))
    Bsz, KD, L = u.shape
    G, N = B.shape[1], B.shape[2]
    rows = KD // G
    dl = delta + (bias.double()[None, :, None] if bias is not None else 0.0)
    if softplus:
        dl = torch.where(dl <= 20.0, torch.log1p(torch.exp(torch.clamp(dl, max=20.0))), dl)
    Bx = B.repeat_interleave(rows, dim=1)          # (Bsz, KD, N, L): the group's B for each of its rows
    Cx = C.repeat_interleave(rows, dim=1)
    h = torch.zeros(Bsz, KD, N, dtype=torch.float64)
    ys = []
    for t in range(L):
        a = torch.exp(dl[:, :, t, None] * A[None])                        # (Bsz, KD, N)
        h = a * h + Bx[:, :, :, t] * (dl[:, :, t] * u[:, :, t])[:, :, None]
        ys.append((Cx[:, :, :, t] * h).sum(-1))
    y = torch.stack(ys, dim=-1) if L else torch.zeros(Bsz, KD, 0, dtype=torch.float64)
    if D is not None:
        y = y + D.double()[None, :, None] * u
    return y, h


shape = st.tuples(st.integers(1, 2),                                   # batch
                  st.sampled_from([1, 2]),                             # groups
                  st.integers(1, 3),                                   # rows per group
                  st.sampled_from([1, 3, 16]),                         # states
                  st.one_of(st.integers(1, 70), st.sampled_from([255, 256, 257, 300, 513])))  # length (chunk = 256)


@settings(max_examples=40, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
@given(shape=shape, softplus=st.booleans(), has_d=st.booleans(), has_bias=st.booleans(), seed=st.integers(0, 2 ** 16))
def test_oracle_equals_the_float64_restatement(shape, softplus, has_d, has_bias, seed):
    Bsz, G, rows, N, L = shape
    KD = G * rows
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(Bsz, KD, L, generator=g)
    # around the softplus threshold in a few places (Appendix A: the <= 20 branch)
    delta = 0.5 * torch.rand(Bsz, KD, L, generator=g) + 25.0 * (torch.rand(Bsz, KD, L, generator=g) > 0.97)
    A = -0.5 * torch.rand(KD, N, generator=g) - 0.05
    Bm = torch.randn(Bsz, G, N, L, generator=g)
    Cm = torch.randn(Bsz, G, N, L, generator=g)
    D = torch.randn(KD, generator=g) if has_d else None
    bias = 0.5 * torch.rand(KD, generator=g) if has_bias else None
    dout = torch.randn(Bsz, KD, L, generator=g)

    leaves = [t.clone().double().requires_grad_() for t in (u, delta, A, Bm, Cm)]
    Dl = D.clone().double().requires_grad_() if has_d else None
    bl = bias.clone().double().requires_grad_() if has_bias else None
    y, h_last = scan_f64(*leaves, Dl, bl, softplus)
    wanted = [t for t in leaves + [Dl, bl] if t is not None]
    grads = list(torch.autograd.grad(y, wanted, dout.double()))
    ref = dict(zip(["du", "ddelta", "dA", "dB", "dC"], grads[:5]))
    rest = grads[5:]
    if has_d:
        ref["dD"] = rest.pop(0)
    if has_bias:
        ref["dbias"] = rest.pop(0)

    out, x = oss_oracle.scan_fwd(u, delta, A, Bm, Cm, D, bias, softplus, chunk=256)
    assert_close(out, y.float(), 2e-5, 2e-5, "out")
    assert_close(x[:, :, -1, 1::2], h_last.float(), 2e-5, 2e-5, "last state")
    got = oss_oracle.scan_bwd(u, delta, A, Bm, Cm, D, bias, dout, x if L > 256 else None, softplus)
    names = ["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"]
    for name, t in zip(names, got):
        if name not in ref:
            continue
        want = ref[name].float()
        sc = max(float(want.abs().max()), 1.0)
        assert_close(t, want, 1e-4, 2e-5 * sc, name)


@pytest.mark.parametrize("L", [0, 1])
def test_degenerate_lengths(L):
    """an empty sequence is legal (outputs empty, weight gradients zero); one step has no recurrence at all"""
    torch.manual_seed(0)
    u, delta = torch.randn(1, 2, L), 0.5 * torch.rand(1, 2, L)
    A, Bm, Cm = -torch.rand(2, 3), torch.randn(1, 1, 3, L), torch.randn(1, 1, 3, L)
    out, x = oss_oracle.scan_fwd(u, delta, A, Bm, Cm, None, None, True, chunk=256)
    y, _ = scan_f64(u, delta, A, Bm, Cm, None, None, True)
    assert tuple(out.shape) == (1, 2, L)
    if L:
        assert_close(out, y.float(), 2e-5, 2e-5, "out")
