"""Training-path plumbing on the MI355X: the flat parameter buffer behind the nn.Parameters, sn_effective_lambdas,
sn_param_penalty, and whole-step hipGraph capture."""
import copy
import io

import numpy as np
import pytest
import torch

import scene_net_amd as sna
from scene_net_amd import _hip
from oracle import geneo_oracle as go
from oracle import loss_oracle as lo

pytestmark = pytest.mark.gpu


def _make(dev, seed=3, ks=(9, 5, 5)):
    torch.manual_seed(seed)
    model = sna.SceneNet({"cy": 2, "cone": 1, "neg": 1}, ks).to(dev)
    with torch.no_grad():
        for n in model.geneos:
            model.lambdas_dict[f"lambda_{n}"].mul_(0.1)
    return model


def _data(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand((2, 1, 12, 12, 16), generator=g) < 0.2).to(dev)
    y = (torch.rand((2, 1, 12, 12, 16), generator=g) < 0.05).to(dev)
    return x, y


def _aliases(model):
    flat = model._flat
    return all(p.data_ptr() == flat.data_ptr() + 4 * i for p, i in model._leaf_slots())


def test_effective_lambdas_kernel_is_bit_exact(hip_device):
    rng = np.random.default_rng(0)
    for G in (1, 3, 16, 33):
        names = [f"{k}_{i}" for k in ("cy", "cone", "neg") for i in range(G)][:G]
        lam = rng.uniform(-0.3, 0.4, G).astype(np.float32)
        last = int(rng.integers(0, G))
        order = sorted(range(G), key=lambda i: f"lambda_{names[i]}")
        want = go.effective_lambdas(lam, last, names)
        buf = torch.from_numpy(lam.copy()).to(hip_device)
        got = _hip.effective_lambdas(buf, torch.tensor(order, dtype=torch.int32, device=hip_device), last)
        assert torch.equal(got.cpu(), want)
        assert buf[last].item() == want[last].item()          # refreshed in place (SCENE_Net.py:333)
        keep = [i for i in range(G) if i != last]
        assert np.array_equal(buf.cpu().numpy()[keep], lam[keep])


def test_param_penalty_kernel_matches_oracle(hip_device):
    rng = np.random.default_rng(1)
    for trial in range(4):
        n_par, n_lam = 11, 5
        vals = rng.uniform(-1.0, 1.0, n_par + n_lam).astype(np.float32)
        if trial == 1:
            vals[n_par:] = np.abs(vals[n_par:]) + 0.5          # sum of free coefficients > 1: last one negative
        last = n_par + 2
        mask = np.array([1] * n_par + [2] * n_lam, dtype=np.int8)
        mask[last] = 0
        P = torch.from_numpy(vals).to(hip_device)
        value, grad = _hip.param_penalty(P, torch.from_numpy(mask).to(hip_device), 1.5, True)
        leaves = [torch.tensor(float(v), requires_grad=True) for v in vals]
        cvx = {f"l{i}": leaves[n_par + i] for i in range(n_lam)}
        cvx[f"l{2}"] = (1 - sum(v for k, v in cvx.items() if k != "l2")).detach()   # frozen last coefficient
        params = {f"p{i}": leaves[i] for i in range(n_par)}
        ref = lo.cvx_loss(cvx, 1.5) + lo.positive_regularizer(params, 1.5)
        ref.backward()
        assert abs(value.item() - ref.item()) <= 1e-6 * max(1.0, abs(ref.item()))
        want = np.array([0.0 if (l.grad is None) else l.grad.item() for l in leaves], dtype=np.float32)
        want[last] = 0.0
        assert np.allclose(grad.cpu().numpy(), want, atol=1e-6)


def test_parameters_alias_one_buffer_and_survive_everything(hip_device):
    model = _make(hip_device)
    x, y = _data(hip_device)
    keys_before = list(model.state_dict().keys())
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    crit = sna.GENEO_Tversky_Loss(targets=y.float(), weighting_scheme_path=None, save_weighting_scheme=False)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = crit(model(x), y, model.get_cvx_coefficients(), model.get_geneo_params())
        loss.backward()
        opt.step()
        return loss.item()

    l0 = step()
    assert _aliases(model)
    flat_id = model._flat.data_ptr()
    for _ in range(3):
        step()
    assert model._flat.data_ptr() == flat_id and _aliases(model)        # optimiser steps update the buffer in place
    # the packed view the kernels read IS the parameters
    r = model.geneos["cy_0"].geneo_params["radius"]
    assert model._flat[_hip.SN_P_RADIUS].item() == r.item()
    # inference in between re-creates the frozen coefficient (SCENE_Net.py:333); the next training forward re-aliases
    with torch.no_grad():
        out_inf = model(x)
    out_trn = model(x)
    assert _aliases(model)
    assert torch.equal(out_inf, out_trn.detach())                       # both paths: same numbers bit for bit
    assert abs(sum(float(p.detach()) for p in model.lambdas_dict.values()) - 1.0) < 1e-5
    # state dict: same keys, loadable into a fresh module, values equal
    assert list(model.state_dict().keys()) == keys_before
    buf = io.BytesIO()
    torch.save(model.state_dict(), buf)
    buf.seek(0)
    fresh = sna.SceneNet({"cy": 2, "cone": 1, "neg": 1}, (9, 5, 5))
    fresh.last_lambda = model.last_lambda
    fresh.load_state_dict(torch.load(buf))
    fresh = fresh.to(hip_device)
    with torch.no_grad():
        assert torch.equal(fresh(x), model(x))
    # deepcopy works and is independent
    twin = copy.deepcopy(model)
    with torch.no_grad():
        twin.geneos["cy_0"].geneo_params["radius"].add_(1.0)
    assert twin.geneos["cy_0"].geneo_params["radius"].item() != model.geneos["cy_0"].geneo_params["radius"].item()
    step()
    assert _aliases(model)
    # moving the module away and back re-establishes the aliasing
    model.cpu()
    model.to(hip_device)
    l1 = step()
    assert _aliases(model) and np.isfinite(l1) and l1 < l0


def test_penalties_take_the_fused_path_and_match_the_generic_one(hip_device):
    model = _make(hip_device, seed=5)
    with torch.no_grad():
        model.geneos["cy_1"].geneo_params["sigma"].fill_(-0.7)
        model.geneos["neg_0"].geneo_params["radius"].fill_(-0.1)
    x, y = _data(hip_device)
    crit = sna.GENEO_Loss(targets=y.float(), weighting_scheme_path=None, save_weighting_scheme=False, convex_weight=2.0)
    out = model(x)
    cvx, gp = model.get_cvx_coefficients(), model.get_geneo_params()
    assert getattr(cvx, "_sn_live").current() is not None
    fused = crit.cvx_loss(cvx) + crit.positive_regularizer(gp)
    both = crit._penalties(cvx, gp)
    generic = crit.cvx_loss({k: v for k, v in cvx.items()}) + crit.positive_regularizer({k: v for k, v in gp.items()})
    assert abs(fused.item() - generic.item()) < 1e-6 and abs(both.item() - generic.item()) < 1e-6
    g_fused = torch.autograd.grad(both, [p for p in model.parameters() if p.requires_grad], retain_graph=True,
                                  allow_unused=True)
    g_gen = torch.autograd.grad(generic, [p for p in model.parameters() if p.requires_grad], allow_unused=True)
    for a, b in zip(g_fused, g_gen):
        assert (0.0 if a is None else a.item()) == pytest.approx(0.0 if b is None else b.item(), abs=1e-6)
    del out


def test_whole_training_step_replays_from_a_hip_graph(hip_device):
    """voxel grid -> forward -> criterion -> backward -> SGD captured once, replayed; same parameters as eager."""
    x, y = _data(hip_device, seed=4)

    def build():
        model = _make(hip_device, seed=7)
        opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=5e-2)
        crit = sna.GENEO_Tversky_Loss(targets=y.float(), weighting_scheme_path=None, save_weighting_scheme=False)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = crit(model(x), y, model.get_cvx_coefficients(), model.get_geneo_params())
            loss.backward()
            opt.step()
            return loss
        return model, step

    eager_model, eager_step = build()
    for _ in range(3 + 4):   # capture itself executes nothing
        eager_step()
    graph_model, graph_step = build()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            graph_step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        graph_step()
    for _ in range(4):
        g.replay()
    torch.cuda.synchronize()
    for (n, a), (_, b) in zip(eager_model.named_parameters(), graph_model.named_parameters()):
        assert torch.equal(a.detach(), b.detach()), n


def test_captured_training_step_api(hip_device):
    """sna.CapturedTrainingStep: points -> grids (+GT) -> forward -> criterion -> backward -> Adam, one graph; the
    warm-up steps it runs and its replays equal the same number of eager steps bit for bit."""
    from scene_net_amd.synthetic import synthetic_tile
    tiles, labels = zip(*[synthetic_tile(t, 20_000) for t in range(2)])

    def build():
        torch.manual_seed(21)
        model = sna.SceneNet({"cy": 2, "cone": 1, "neg": 1}, (9, 5, 5)).to(hip_device)
        with torch.no_grad():
            for n in model.geneos:
                model.lambdas_dict[f"lambda_{n}"].mul_(0.1)
        batch = sna.PointBatch.from_tiles(tiles, labels, device=hip_device)
        pipe = sna.ScenePipeline(model, (32, 32, 32), keep_labels=[15.0])
        gt = pipe.voxelize(batch, want_gt=True).gt_occ
        crit = sna.GENEO_Tversky_Loss(targets=gt.float(), weighting_scheme_path=None, save_weighting_scheme=False)
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2, capturable=True)
        return model, batch, pipe, crit, opt

    model, batch, pipe, crit, opt = build()
    cap = sna.CapturedTrainingStep(pipe, crit, opt, batch, warmup=2)
    losses = [cap.replay().item() for _ in range(5)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    model_e, batch_e, pipe_e, crit_e, opt_e = build()
    for _ in range(2 + 5):
        opt_e.zero_grad(set_to_none=True)
        g = pipe_e.voxelize(batch_e, want_gt=True)
        loss_e = crit_e(model_e(g.occ), g.gt_occ, model_e.get_cvx_coefficients(), model_e.get_geneo_params())
        loss_e.backward()
        opt_e.step()
    assert loss_e.item() == losses[-1]
    for (n, a), (_, b) in zip(model_e.named_parameters(), model.named_parameters()):
        assert torch.equal(a.detach(), b.detach()), n


def test_training_step_with_the_opener_riding_in_the_voxelisation(hip_device):
    """9 x 9 x 9 bank: training.voxelize_and_forward hands the forward's opener (bank + effective coefficients) to the
    voxelisation's first launch (SceneNet.train_rider -> sn_voxel_occupancy_fused_bank with lambdas) -- the same bank, the
    same coefficients, the same refreshed frozen coefficient, hence the same losses and parameters, bit for bit, as the
    step whose forward opens with sn_geneo_bank_lambdas; eagerly and as a CapturedTrainingStep."""
    from scene_net_amd.synthetic import synthetic_tile
    from scene_net_amd.training import voxelize_and_forward
    tiles, labels = zip(*[synthetic_tile(t, 20_000) for t in range(2)])

    def build():
        torch.manual_seed(5)
        model = sna.SceneNet({"cy": 2, "cone": 2, "neg": 1}, (9, 9, 9)).to(hip_device)
        with torch.no_grad():
            for n in model.geneos:
                model.lambdas_dict[f"lambda_{n}"].mul_(0.1)
        batch = sna.PointBatch.from_tiles(tiles, labels, device=hip_device)
        pipe = sna.ScenePipeline(model, (32, 32, 32), keep_labels=[15.0])
        gt = pipe.voxelize(batch, want_gt=True).gt_occ
        crit = sna.GENEO_Tversky_Loss(targets=gt.float(), weighting_scheme_path=None, save_weighting_scheme=False)
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2, capturable=True)
        return model, batch, pipe, crit, opt

    # the rider's outputs against the forward's own opener, on the same parameters
    model, batch, pipe, crit, opt = build()
    assert pipe.rides(2)
    flat, meta, _ = model._flat_sync(hip_device)
    n = meta["G"] * _hip.SN_NPARAM
    lam_before = flat[n:].clone()
    bank_o, lam_o = _hip.geneo_bank_lambdas(flat[:n].view(meta["G"], _hip.SN_NPARAM), meta["kinds"], (9, 9, 9),
                                            flat[n:].clone(), meta["order"], meta["last"])
    rider, (bank_r, lam_r) = model.train_rider(hip_device)
    grids = pipe.voxelize(batch, want_gt=True, bank_rider=rider)
    assert grids.rider_done and torch.equal(bank_r, bank_o) and torch.equal(lam_r, lam_o)
    assert torch.equal(flat[n:], torch.where(torch.arange(meta["G"], device=hip_device) == meta["last"], lam_o, lam_before))

    def serial_steps(k):
        model_e, batch_e, pipe_e, crit_e, opt_e = build()
        for _ in range(k):
            opt_e.zero_grad(set_to_none=True)
            g = pipe_e.voxelize(batch_e, want_gt=True)
            loss = crit_e(model_e(g.occ), g.gt_occ, model_e.get_cvx_coefficients(), model_e.get_geneo_params())
            loss.backward()
            opt_e.step()
        return model_e, loss

    model_r, batch_r, pipe_r, crit_r, opt_r = build()
    for _ in range(4):
        opt_r.zero_grad(set_to_none=True)
        g, pred = voxelize_and_forward(pipe_r, batch_r)
        loss_r = crit_r(pred, g.gt_occ, model_r.get_cvx_coefficients(), model_r.get_geneo_params())
        loss_r.backward()
        opt_r.step()
    model_e, loss_e = serial_steps(4)
    assert loss_r.item() == loss_e.item()
    for (nm, a), (_, b) in zip(model_e.named_parameters(), model_r.named_parameters()):
        assert torch.equal(a.detach(), b.detach()), nm

    model_c, batch_c, pipe_c, crit_c, opt_c = build()
    cap = sna.CapturedTrainingStep(pipe_c, crit_c, opt_c, batch_c, warmup=2)
    losses = [cap.replay().item() for _ in range(3)]
    model_e, loss_e = serial_steps(2 + 3)
    assert loss_e.item() == losses[-1]
    for (nm, a), (_, b) in zip(model_e.named_parameters(), model_c.named_parameters()):
        assert torch.equal(a.detach(), b.detach()), nm

    # a validation pass BETWEEN replays runs on the replayed weights.  The replayed optimiser graph writes the
    # parameters through raw pointers -- no `_version` moves -- so replay() drops the model's parameter-derived caches
    # (packed parameters, coefficients, K3L tables, learnt verdicts); without that the second pass below would reuse the
    # first one's.  Reference: a freshly built model loaded with the same parameter values.
    def fresh_copy():
        m = sna.SceneNet({"cy": 2, "cone": 2, "neg": 1}, (9, 9, 9)).to(hip_device)
        m.last_lambda = model_c.last_lambda
        m.load_state_dict({k: v.detach().clone() for k, v in model_c.state_dict().items()})
        return sna.ScenePipeline(m, (32, 32, 32), keep_labels=[15.0])

    with torch.no_grad():
        for fused in (True, False):
            model_c.fused_forward = fused
            seen = []
            for _ in range(2):
                val = pipe_c(batch_c)
                torch.cuda.synchronize()
                val = pipe_c(batch_c)                 # (second call: whatever the first one cached is now in use)
                ref_pipe = fresh_copy()
                ref_pipe.model.fused_forward = fused
                assert torch.equal(val, ref_pipe(batch_c))
                seen.append(val.clone())
                for _ in range(2):
                    cap.replay()
            assert not torch.equal(seen[0], seen[1])      # the weights did move in between


@pytest.mark.parametrize("crit_cls", [sna.GENEO_Tversky_Loss, sna.GENEO_Dice_Loss, sna.GENEO_Loss])
@pytest.mark.parametrize("bf16", [False, True])
def test_fused_criterion_equals_its_parts(hip_device, crit_cls, bf16):
    """GENEO_Loss.forward on a live scene_net_amd model: dense terms + penalties as ONE op in the dense loss's launches
    (sn_criterion_forward / _backward) == the dense op + the penalty op + torch's add (and, backward, its multiply): the loss
    and every parameter's gradient bit for bit, for float32 and bf16 predictions, with an upstream factor."""
    x, y = _data(hip_device, seed=9)

    def run(fused):
        model = _make(hip_device, seed=13)
        if bf16:
            model.activation_dtype = torch.bfloat16
        with torch.no_grad():   # a negative coefficient and a negative parameter: both penalties are active
            next(iter(model.lambdas_dict.values())).fill_(-0.3)
            next(iter(model.geneos.values())).geneo_params["sigma"].fill_(-0.2)
        crit = crit_cls(targets=y.float(), weighting_scheme_path=None, save_weighting_scheme=False)
        pred = model(x)
        cvx, par = model.get_cvx_coefficients(), model.get_geneo_params()
        if fused:
            loss = crit(pred, y, cvx, par)
        else:
            terms, cfg = crit._terms()
            loss = crit._dense(pred, y, terms, **cfg) + crit._penalties(cvx, par)
        (loss * 1.75).backward()
        return loss.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    loss_f, grads_f = run(True)
    loss_p, grads_p = run(False)
    assert loss_f.dtype == loss_p.dtype == torch.float32 and torch.equal(loss_f, loss_p)
    assert grads_f.keys() == grads_p.keys() and len(grads_f) > 10
    for n in grads_f:
        assert torch.equal(grads_f[n], grads_p[n]), n


def test_rider_buffers_are_guarded_against_a_later_forward(hip_device):
    """train_rider()'s (bank, lam) live in persistent buffers the NEXT rider launch rewrites in place (no version
    bump for autograd to see).  Two sets are used alternately: one other forward may run before a backward (same gradients
    as without it); after a second one the first graph's backward raises instead of reading another step's bank."""
    from scene_net_amd.synthetic import synthetic_tile
    from scene_net_amd.training import voxelize_and_forward
    tiles, labels = zip(*[synthetic_tile(t, 20_000) for t in range(2)])
    torch.manual_seed(3)
    model = sna.SceneNet({"cy": 2, "cone": 1, "neg": 1}, (9, 9, 9)).to(hip_device)
    batch = sna.PointBatch.from_tiles(tiles, labels, device=hip_device)
    pipe = sna.ScenePipeline(model, (32, 32, 32), keep_labels=[15.0])
    assert pipe.rides(2)

    def grads_after(n_other_forwards):
        model.zero_grad(set_to_none=True)
        g, out = voxelize_and_forward(pipe, batch)
        keep = []
        for _ in range(n_other_forwards):
            with torch.no_grad():
                next(iter(model.geneos.values())).geneo_params["radius"].add_(0.01)   # the other steps see other parameters
            keep.append(voxelize_and_forward(pipe, batch))
        out.float().sum().backward()
        return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    g0 = grads_after(0)
    with torch.no_grad():
        next(iter(model.geneos.values())).geneo_params["radius"].sub_(0.0)
    torch.manual_seed(3)
    model2 = sna.SceneNet({"cy": 2, "cone": 1, "neg": 1}, (9, 9, 9)).to(hip_device)
    pipe2 = sna.ScenePipeline(model2, (32, 32, 32), keep_labels=[15.0])
    model, pipe = model2, pipe2
    g1 = grads_after(1)          # one overlapping forward (with changed parameters): the first graph still reads ITS bank
    assert set(g0) == set(g1) and all(torch.equal(g0[n], g1[n]) for n in g0)
    with pytest.raises(RuntimeError, match="handed to a later forward"):
        grads_after(2)
