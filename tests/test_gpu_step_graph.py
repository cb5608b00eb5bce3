"""The whole training iteration replayed as one HIP graph (train.Trainer(graph=True)) against the same iteration
enqueued launch by launch: the same kernels on the same data, so parameters, optimizer state and losses are compared
BIT FOR BIT, with the random draws injected (t, u, noise are graph inputs) and with torch's device RNG."""
import copy

import pytest
import torch

from conftest import SMALL, TINY

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair(hp, seed=3):
    from view_fusion_amd import train
    a = train.build_model(unet_params=hp, device=DEV, seed=seed)
    b = copy.deepcopy(a)
    return a, b


def _batches(n, B, N, hw, ragged=False):
    from view_fusion_amd import train
    return [train.synthetic_batch(B, N, hw, device=DEV, seed=100 + i, ragged=ragged) for i in range(n)]


def _draws(i, B, hw, T=2000):
    g = torch.Generator().manual_seed(900 + i)
    return dict(t=torch.randint(1, T, (B,), generator=g).to(DEV), u=torch.rand(B, 1, generator=g).to(DEV),
                noise=torch.randn(B, 3, hw, hw, generator=g).to(DEV))


def _same(ma, mb):
    for (k, p), q in zip(ma.named_parameters(), mb.parameters()):
        assert torch.equal(p, q), k


@pytest.mark.parametrize("hp,B,N,hw,n", [(TINY, 2, 3, 16, 6), (SMALL, 4, 2, 64, 6),
                                         # the geometries bench.py times: BASELINE C2 (S = 96: Winograd tail-split
                                         # workspaces, wino_pack_multi, the deferred colsum_multi table inside the
                                         # capture) and C4 (S = 48), three replays each
                                         (SMALL, 16, 6, 64, 5), (SMALL, 8, 6, 64, 5)])
def test_graph_step_matches_eager_bitwise_with_injected_draws(hp, B, N, hw, n):
    from view_fusion_amd import train
    ma, mb = _pair(hp)
    ta, tb = train.Trainer(ma, graph=False, lr_warmup=4), train.Trainer(mb, graph=True, lr_warmup=4)
    batches = _batches(n, B, N, hw)
    vc = batches[0]["view_count"]
    for i, bt in enumerate(batches):
        bt["view_count"] = vc                        # one geometry
        la, lb = ta.step(bt, **_draws(i, B, hw)), tb.step(bt, **_draws(i, B, hw))
        assert torch.equal(la, lb), i
        _same(ma, mb)
    assert ta.graph_steps == 0 and tb.graph_steps == n - train.Trainer.GRAPH_AFTER
    # .grad shows the last iteration's gradients, the optimizer state continues where the eager one is
    for p, q in zip(ma.parameters(), mb.parameters()):
        assert torch.equal(p.grad, q.grad)
    sa, sb = ta.opt.state_dict()["state"], tb.opt.state_dict()["state"]
    for k in sa:
        assert float(sa[k]["step"]) == float(sb[k]["step"]) == n
        assert torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"]) and torch.equal(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"])
    # an eager iteration after the replays (a kernel log forces it) continues from the same state
    bt = batches[0]
    la, lb = ta.step(bt, **_draws(9, B, hw)), tb.step(bt, **_draws(9, B, hw), y_t=None)   # y_t: not a graph input
    assert torch.equal(la, lb)
    _same(ma, mb)


def test_graph_step_with_device_rng_matches_eager():
    """No injection: the captured draws (randint, rand, randn) advance torch's Philox offset per replay exactly as
    the eager launches do, so two runs from the same seed stay identical."""
    from view_fusion_amd import train
    ma, mb = _pair(TINY)
    ta, tb = train.Trainer(ma, graph=False), train.Trainer(mb, graph=True)
    bt = _batches(1, 2, 3, 16)[0]
    la, lb = [], []
    torch.manual_seed(11)
    for _ in range(5):
        la.append(ta.step(bt))
    torch.manual_seed(11)
    for _ in range(5):
        lb.append(tb.step(bt))
    assert tb.graph_steps == 3
    assert len({float(x) for x in lb}) == 5           # every replay drew new (t, u, noise)
    assert all(torch.equal(x, y) for x, y in zip(la, lb))
    _same(ma, mb)


def test_graph_step_two_geometries_and_ragged_fallback():
    from view_fusion_amd import train
    ma, mb = _pair(TINY)
    ta, tb = train.Trainer(ma, graph=False), train.Trainer(mb, graph=True)
    b1, b2 = _batches(2, 2, 3, 16)
    b2["view_count"] = torch.tensor([1, 2])
    order = [b1, b1, b2, b2, b1, b2, b1, b2, b2, b1]
    for i, bt in enumerate(order):
        la, lb = ta.step(bt, **_draws(i, 2, 16)), tb.step(bt, **_draws(i, 2, 16))
        assert torch.equal(la, lb), i
        for p, q in zip(ma.parameters(), mb.parameters()):
            assert torch.equal(p, q) and torch.equal(p.grad, q.grad)
    assert tb.graph_steps == 6 and len(tb._graphs) == 2
    # a device-resident view_count is never read back per step: eager
    b3 = dict(b1, view_count=b1["view_count"].to(DEV))
    n = tb.graph_steps
    assert torch.equal(ta.step(b3, **_draws(20, 2, 16)), tb.step(b3, **_draws(20, 2, 16)))
    assert tb.graph_steps == n
    _same(ma, mb)


def test_graph_step_ragged_view_counts_share_one_graph_per_stacked_count():
    """The reference draws view_count per sample and iteration: vectors with the same sum replay the same graph, with
    the offsets table as an input."""
    from view_fusion_amd import train
    ma, mb = _pair(TINY)
    ta, tb = train.Trainer(ma, graph=False), train.Trainer(mb, graph=True)
    vcs = [[1, 2, 3], [3, 2, 1], [2, 2, 2], [3, 1, 2], [2, 3, 1], [2, 2, 2], [1, 1, 1], [1, 2, 3]]
    for i, (bt, vc) in enumerate(zip(_batches(len(vcs), 3, 3, 16), vcs)):
        bt["view_count"] = torch.tensor(vc)
        la, lb = ta.step(bt, **_draws(i, 3, 16)), tb.step(bt, **_draws(i, 3, 16))
        assert torch.equal(la, lb), i
        for p, q in zip(ma.parameters(), mb.parameters()):
            assert torch.equal(p, q) and torch.equal(p.grad, q.grad), i
    assert tb.graph_steps == 5 and len(tb._graphs) == 2        # S = 6 (captured), S = 3 (seen once)
    bad = dict(_batches(1, 3, 3, 16)[0], view_count=torch.tensor([1, 0, 3]))
    with pytest.raises(ValueError):
        tb.step(bad)


def test_sampling_after_graph_steps_sees_updated_weights():
    """The packed-weight caches of the inference path are keyed on the parameters' version counters, which every
    replay bumps: a sampler call after replays must not reuse packs made before them."""
    from view_fusion_amd import train
    ma, mb = _pair(TINY)
    ta, tb = train.Trainer(ma, graph=False, lr_warmup=2), train.Trainer(mb, graph=True, lr_warmup=2)
    bt = _batches(1, 2, 3, 16)[0]
    y_t = torch.randn(2, 3, 16, 16, device=DEV)
    t = torch.tensor([7, 3], device=DEV)
    z = torch.randn(2, 3, 16, 16, device=DEV)
    for i in range(5):
        ta.step(bt, **_draws(i, 2, 16)), tb.step(bt, **_draws(i, 2, 16))
        ya = ma.p_sample(y_t, bt["y_cond"], bt["view_count"], bt["angle"], t, z=z)[0]
        yb = mb.p_sample(y_t, bt["y_cond"], bt["view_count"], bt["angle"], t, z=z)[0]
        assert torch.equal(ya, yb), i
    assert tb.graph_steps == 3


def test_graph_steps_survive_an_optimizer_state_reload():
    """load_state_dict replaces the Adam moment tensors: captured steps that address the old ones are dropped and the
    geometry is captured again -- the run continues bit-identically to an eager one that reloads the same state."""
    from view_fusion_amd import train
    ma, mb = _pair(TINY)
    ta, tb = train.Trainer(ma, graph=False, lr_warmup=3), train.Trainer(mb, graph=True, lr_warmup=3)
    bt = _batches(1, 2, 3, 16)[0]
    for i in range(4):
        assert torch.equal(ta.step(bt, **_draws(i, 2, 16)), tb.step(bt, **_draws(i, 2, 16)))
    assert tb.graph_steps == 2
    for tr in (ta, tb):
        sd = copy.deepcopy(tr.opt.state_dict())
        for st in sd["state"].values():                 # a state that differs from the live one
            st["exp_avg"] = st["exp_avg"] * 0.5
        tr.opt.load_state_dict(sd)
    for i in range(4, 9):
        assert torch.equal(ta.step(bt, **_draws(i, 2, 16)), tb.step(bt, **_draws(i, 2, 16))), i
        _same(ma, mb)
    assert tb.graph_steps == 2 + 3                      # two eager sightings after the reload, then replays again
    sa, sb = ta.opt.state_dict()["state"], tb.opt.state_dict()["state"]
    for k in sa:
        assert float(sa[k]["step"]) == float(sb[k]["step"]) == 9 and torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"])


def test_graph_capture_with_an_autograd_graph_of_the_model_kept_alive():
    """A caller that keeps a tensor with autograd history of the model (a loss it never backpropagated) keeps the
    parameters' AccumulateGrad nodes bound to the default stream; the captured step must not route through them."""
    from view_fusion_amd import train
    ma, mb = _pair(TINY)
    ta, tb = train.Trainer(ma, graph=False), train.Trainer(mb, graph=True)
    bt = _batches(1, 2, 3, 16)[0]
    kept = [m(y_0=bt["y_0"], y_cond=bt["y_cond"], view_count=bt["view_count"], angle=bt["angle"]) for m in (ma, mb)]
    assert all(k.requires_grad for k in kept)
    for i in range(5):
        assert torch.equal(ta.step(bt, **_draws(i, 2, 16)), tb.step(bt, **_draws(i, 2, 16)))
        for p, q in zip(ma.parameters(), mb.parameters()):
            assert torch.equal(p, q) and torch.equal(p.grad, q.grad)
    assert tb.graph_steps == 3 and all(k.requires_grad for k in kept)
