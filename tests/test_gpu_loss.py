"""K5 on the MI355X: sn_loss_forward / sn_loss_backward behind the reference's criterion classes, against
 (1) losses and gradients the reference's own classes produced (tests/golden/geneo_loss.npz), and
 (2) the pinned oracle (oracle/loss_oracle.py) on larger seeded inputs.
Tolerance (floating point, stated here): the kernels accumulate in fp64 where the reference reduces in the tensor
dtype, so fp32 cases agree to 2e-5 relative (loss) / 2e-5 of the largest gradient entry.  fp64 cases agree to 1e-6
only, because the reference's weights are fp32 whatever the input dtype (int64/int64 division, w_mse.py:130) and its
`weights / mean(weights)` is an fp32 reduction over all elements (w_mse.py:144); the criteria without weights
(Tversky, dice) agree to 1e-12."""
import ast
import os

import numpy as np
import pytest
import torch

import scene_net_amd as sna
from scene_net_amd import _hip
from oracle import loss_oracle as lo

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "geneo_loss.npz"))
CASES = [str(c) for c in GOLD["cases"]]
CLASSES = {"geneo": sna.GENEO_Loss, "tversky": sna.GENEO_Tversky_Loss, "dice": sna.GENEO_Dice_Loss}


def _tol(dtype):
    return 2e-5 if dtype == torch.float32 else 1e-6


def _build(name, kind, dev):
    hp = ast.literal_eval(str(GOLD[f"{name}|hp"]))
    if kind != "tversky":
        hp = {k: v for k, v in hp.items() if not k.startswith(("tversky", "focal"))}
    gt = torch.from_numpy(GOLD[f"{name}|gt"])
    crit = CLASSES[kind](targets=gt, weighting_scheme_path=None, save_weighting_scheme=False, **hp)
    crit.freqs = torch.from_numpy(GOLD[f"{name}|freqs"]).to(dev)
    names = [str(n) for n in GOLD[f"{name}|cvx_names"]]
    cvx = torch.nn.ParameterDict({n: torch.nn.Parameter(torch.tensor(float(v), device=dev), requires_grad=(n != names[-1]))
                                  for n, v in zip(names, GOLD[f"{name}|cvx_values"])})
    gp = torch.nn.ParameterDict({str(n): torch.nn.Parameter(torch.tensor(float(v), device=dev))
                                 for n, v in zip(GOLD[f"{name}|param_names"], GOLD[f"{name}|param_values"])})
    return crit, gt, cvx, gp, names


@pytest.mark.parametrize("kind", list(CLASSES))
@pytest.mark.parametrize("name", CASES)
def test_reference_golden_losses_and_gradients(hip_device, name, kind):
    crit, gt, cvx, gp, names = _build(name, kind, hip_device)
    assert torch.equal(crit.hist_frequency_estimation(gt.flatten())[0].cpu(), torch.from_numpy(GOLD[f"{name}|est_freqs"]))
    pred = torch.from_numpy(GOLD[f"{name}|pred"]).to(hip_device).requires_grad_(True)
    loss = crit(pred, gt.to(hip_device), cvx, gp)
    loss.backward()
    ref = GOLD[f"{name}|{kind}|loss"]
    tol = _tol(pred.dtype)
    assert loss.dtype == pred.dtype
    assert abs(loss.item() - float(ref)) <= tol * abs(float(ref)), (loss.item(), float(ref))
    gref = GOLD[f"{name}|{kind}|grad_pred"]
    err = np.abs(pred.grad.cpu().numpy() - gref).max()
    assert err <= tol * np.abs(gref).max(), (err, np.abs(gref).max())
    got = np.array([0.0 if cvx[n].grad is None else cvx[n].grad.item() for n in names])
    np.testing.assert_allclose(got, GOLD[f"{name}|{kind}|grad_cvx"], rtol=1e-6, atol=0)
    got = np.array([p.grad.item() for p in gp.values()])
    np.testing.assert_allclose(got, GOLD[f"{name}|{kind}|grad_params"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("red", ["mean", "sum"])
@pytest.mark.parametrize("name", CASES)
def test_dice_bce_reference_golden(hip_device, name, red):
    """BinaryDiceLoss_BCE ('dice_bce'): weights * BCELoss + dice, against the reference's own numbers."""
    hp = ast.literal_eval(str(GOLD[f"{name}|hp"]))
    hp = {k: v for k, v in hp.items() if k in ("weight_alpha", "weight_epsilon", "mse_weight")}
    gt = torch.from_numpy(GOLD[f"{name}|gt"])
    crit = sna.BinaryDiceLoss_BCE(targets=gt, weighting_scheme_path=None, save_weighting_scheme=False, reduction=red, **hp)
    crit.freqs = torch.from_numpy(GOLD[f"{name}|freqs"]).to(hip_device)
    pred = torch.from_numpy(GOLD[f"{name}|pred"]).clamp(1e-4, 1 - 1e-4).to(hip_device).requires_grad_(True)
    loss = crit(pred, gt.to(hip_device))
    loss.backward()
    ref, gref = float(GOLD[f"{name}|dice_bce_{red}|loss"]), GOLD[f"{name}|dice_bce_{red}|grad_pred"]
    tol = _tol(pred.dtype)
    assert abs(loss.item() - ref) <= tol * abs(ref), (loss.item(), ref)
    err = np.abs(pred.grad.cpu().numpy() - gref).max()
    assert err <= tol * np.abs(gref).max(), (err, np.abs(gref).max())


def test_bce_clamps_like_torch(hip_device):
    """p = 0 and p = 1 exactly: log clamped at -100, gradient denominator at 1e-12 (torch's BCELoss)."""
    pred = torch.tensor([[0.0, 1.0, 0.0, 1.0, 0.3, 0.999999]], dtype=torch.float64)
    gt = torch.tensor([[0.0, 1.0, 1.0, 0.0, 1.0, 0.0]], dtype=torch.float64)
    ranges = torch.zeros(1, dtype=torch.float32, device=hip_device)
    bin_w = torch.ones(1, dtype=torch.float32, device=hip_device)
    loss, _, coef, _ = _hip.loss_forward(pred.to(hip_device), gt.to(hip_device), ranges, bin_w, _hip.SN_LOSS_WBCE)
    po = pred.clone().requires_grad_(True)
    ref = torch.nn.functional.binary_cross_entropy(po, gt)
    ref.backward()
    assert abs(loss[0].item() - ref.item()) <= 1e-12 * ref.item()
    g = _hip.loss_backward(pred.to(hip_device), gt.to(hip_device), ranges, coef).cpu()
    assert torch.allclose(g, po.grad, rtol=1e-12, atol=0)


@pytest.mark.parametrize("name", CASES)
def test_weight_target_matches_reference(hip_device, name):
    crit, gt, _, _, _ = _build(name, "geneo", hip_device)
    w = crit.get_weight_target(gt.to(hip_device))
    ref = torch.from_numpy(GOLD[f"{name}|weights"])
    assert w.dtype == ref.dtype
    assert (w.cpu() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()  # mean over n elements in fp32


FREQS = [3_000_000, 1200, 800, 700, 650, 400, 300, 310, 150, 9000]


@pytest.mark.parametrize("pred_dt,gt_dt", [(torch.float32, torch.bool), (torch.float32, torch.uint8),
                                           (torch.float32, torch.float32), (torch.float64, torch.float64),
                                           (torch.float32, torch.float64), (torch.float64, torch.float32)])
@pytest.mark.parametrize("shape", [(4, 1, 64, 64, 64), (3, 1, 7, 9, 11), (1, 1, 5, 3, 3), (2, 4099)])
def test_against_oracle_all_dtypes_and_ragged_shapes(hip_device, pred_dt, gt_dt, shape):
    g = torch.Generator().manual_seed(sum(shape))
    u = torch.rand(shape, generator=g, dtype=torch.float64)
    if gt_dt in (torch.bool, torch.uint8):
        gt = (u < 0.05).to(gt_dt)
    else:
        gt = torch.where(u < 0.9, torch.zeros_like(u), torch.where(u < 0.95, torch.ones_like(u), (u - 0.95) * 20)).to(gt_dt)
    pred = torch.rand(shape, generator=g, dtype=torch.float64).to(pred_dt)
    freqs = torch.tensor(FREQS)
    ranges = torch.linspace(0, 1, 11)[:-1]
    hp = dict(alpha=1.5, eps=0.05, mse_weight=2.0)
    # oracle in fp64 on the same values (gt binning happens in gt's own dtype, as the reference would)
    po = pred.detach().clone().double().requires_grad_(True)
    gto = gt.to(torch.float32 if gt_dt in (torch.bool, torch.uint8) else gt_dt)
    w = lo.weight_target(gto, freqs, ranges, hp["alpha"], hp["eps"]).double()
    ref = torch.mean(hp["mse_weight"] * w * (gto.double() - po) ** 2) + \
        lo.focal_tversky_loss(po, gto.double(), 0.3, 0.7, 2.0, 0.5)
    ref.backward()
    crit = sna.GENEO_Tversky_Loss(targets=torch.zeros(4), weighting_scheme_path=None, save_weighting_scheme=False,
                                  weight_alpha=hp["alpha"], weight_epsilon=hp["eps"], mse_weight=hp["mse_weight"],
                                  tversky_alpha=0.3, tversky_beta=0.7, focal_gamma=2.0, tversky_smooth=0.5)
    crit.freqs = freqs.to(hip_device)
    pd = pred.to(hip_device).requires_grad_(True)
    loss = crit(pd, gt.to(hip_device), {}, {})
    loss.backward()
    tol = 5e-6 if pred_dt == torch.float32 else 1e-6   # oracle weights' mean is an fp32 reduction
    assert abs(loss.item() - ref.item()) <= tol * abs(ref.item()), (loss.item(), ref.item())
    err = (pd.grad.cpu().double() - po.grad).abs().max().item()
    assert err <= tol * po.grad.abs().max().item(), (err, po.grad.abs().max().item())


def test_full_size_properties(hip_device):
    """C2 size (32 x 64^3): counts add up, the loss is bit-reproducible, the gradient is linear in the upstream
    scalar and matches a central difference of the loss along a random direction."""
    torch.manual_seed(0)
    B, n = 32, 64 ** 3
    gt = (torch.rand((B, 1, 64, 64, 64), device=hip_device) < 0.02)
    pred = torch.rand((B, 1, 64, 64, 64), device=hip_device, dtype=torch.float64)
    ranges = torch.linspace(0, 1, 11)[:-1].to(hip_device).contiguous()
    bin_w = torch.linspace(0.1, 1.0, 10).to(hip_device).contiguous()
    terms = _hip.SN_LOSS_WMSE | _hip.SN_LOSS_FOCAL_TVERSKY | _hip.SN_LOSS_DICE | _hip.SN_LOSS_WBCE
    pred = pred.clamp(0.05, 0.95)   # keeps the BCE term's higher derivatives tame for the difference quotient
    l1, stats, coef, l1f = _hip.loss_forward(pred, gt, ranges, bin_w, terms, focal_gamma=1.5)
    l2, stats2, coef2, _ = _hip.loss_forward(pred, gt, ranges, bin_w, terms, focal_gamma=1.5)
    assert torch.equal(l1f, l1.float())   # the float32 mirror: the same five numbers rounded once
    assert torch.equal(l1, l2) and torch.equal(stats, stats2) and torch.equal(coef, coef2)
    assert stats[:, :10].sum().item() == B * n
    assert stats[:, 0].sum().item() == (~gt).sum().item() and stats[:, 9].sum().item() == gt.sum().item()
    assert abs(l1[0].item() - (l1[1] + l1[2] + l1[3] + l1[4]).item()) < 1e-12
    g1 = _hip.loss_backward(pred, gt, ranges, coef)
    up = torch.tensor([2.5], dtype=torch.float64, device=hip_device)
    g2 = _hip.loss_backward(pred, gt, ranges, coef, up)
    assert (g2 - 2.5 * g1).abs().max().item() <= 1e-12 * g1.abs().max().item()
    assert torch.equal(_hip.loss_backward(pred, gt, ranges, coef, up.float()), g2)   # a float32 upstream is read as it is
    d = g1 / g1.abs().max() + 0.1 * torch.randn_like(pred)
    h = 1e-5
    lp = _hip.loss_forward(pred + h * d, gt, ranges, bin_w, terms, focal_gamma=1.5)[0][0].item()
    lm = _hip.loss_forward(pred - h * d, gt, ranges, bin_w, terms, focal_gamma=1.5)[0][0].item()
    fd = (lp - lm) / (2 * h)
    an = (g1.double() * d.double()).sum().item()
    assert abs(fd - an) <= 1e-6 * max(abs(fd), abs(an)), (fd, an)


def test_standalone_criteria(hip_device):
    torch.manual_seed(1)
    pred = torch.rand((3, 1, 8, 8, 8), dtype=torch.float64)
    gt = (torch.rand((3, 1, 8, 8, 8)) < 0.2).double()
    for mod, ref in [(sna.TverskyLoss(0.3, 0.7, 2.0), lo.tversky_loss(pred, gt, 0.3, 0.7, 2.0)),
                     (sna.FocalTverskyLoss(0.4, 0.6, 3.0, 1.0), lo.focal_tversky_loss(pred, gt, 0.4, 0.6, 3.0, 1.0)),
                     (sna.BinaryDiceLoss(), lo.binary_dice_loss(pred, gt)),
                     (sna.BinaryDiceLoss(smooth=0.5, reduction="sum"), lo.binary_dice_loss(pred, gt, 0.5, 2, "sum"))]:
        got = mod(pred.to(hip_device), gt.to(hip_device))
        assert abs(got.item() - ref.item()) <= 1e-12 * abs(ref.item()), (type(mod).__name__, got.item(), ref.item())


def test_training_step_through_model_and_criterion(hip_device):
    """forward (HIP) -> GENEO_Tversky_Loss (HIP) -> backward (HIP) == autograd through the oracle's forward + loss."""
    from oracle import geneo_oracle as go
    torch.manual_seed(11)
    ks = (9, 7, 7)
    model = sna.SceneNet({"cy": 2, "cone": 2, "neg": 1}, ks).to(hip_device)
    with torch.no_grad():
        for n in model.geneos:
            model.lambdas_dict[f"lambda_{n}"].mul_(0.05)
        model.geneos["cy_1"].geneo_params["sigma"].fill_(-0.4)   # exercises positive_regularizer
    x = (torch.rand(2, 1, 20, 16, 24) < 0.1)
    gt = (torch.rand(2, 1, 20, 16, 24) < 0.04)
    names = list(model.geneos.keys())
    leaf = {n: {k: p.detach().cpu().clone().requires_grad_(k != "apex") for k, p in model.geneos[n].geneo_params.items()}
            for n in names}
    lam_leaf = [model.lambdas_dict[f"lambda_{n}"].detach().cpu().clone().requires_grad_(True) for n in names]
    last = names.index(model.last_lambda.replace("lambda_", ""))
    lam_leaf[last].requires_grad_(False)
    specs = [(n.split("_")[0], leaf[n]) for n in names]
    po = go.scenenet_forward(x.double(), specs, ks, lam_leaf, last, names=names)
    freqs, ranges = lo.hist_frequency_estimation(gt.float())
    cvx = {f"lambda_{n}": lam_leaf[i] for i, n in enumerate(names)}
    flat = {f"{n}.{k}": v for n in names for k, v in leaf[n].items()}
    ref = lo.geneo_tversky_loss(po, gt.double(), cvx, flat, freqs, ranges, gamma=2.0)
    ref.backward()

    crit = sna.GENEO_Tversky_Loss(targets=gt.float(), weighting_scheme_path=None, save_weighting_scheme=False,
                                  focal_gamma=2.0)
    pred = model(x.to(hip_device))
    loss = crit(pred, gt.to(hip_device), model.get_cvx_coefficients(), model.get_geneo_params())
    loss.backward()
    assert abs(loss.item() - ref.item()) <= 1e-4 * abs(ref.item()), (loss.item(), ref.item())

    def close(a, b):
        a, b = float(a), float(b)
        return abs(a - b) <= 2e-4 + 2e-3 * max(abs(a), abs(b))
    for n in names:
        for k, p in model.geneos[n].geneo_params.items():
            if k != "apex":
                assert close(p.grad, leaf[n][k].grad), (n, k, float(p.grad), float(leaf[n][k].grad))
    for i, n in enumerate(names):
        if i != last:
            assert close(model.lambdas_dict[f"lambda_{n}"].grad, lam_leaf[i].grad), n


def test_cabi_rejects_bad_arguments(hip_device):
    pred = torch.rand((2, 64), device=hip_device)
    gt = torch.rand((2, 64), device=hip_device)
    ranges = torch.linspace(0, 1, 18)[:-1].to(hip_device).contiguous()     # 17 bins > SN_LOSS_MAX_BINS
    with pytest.raises(sna.HipLibraryError, match="bins"):
        _hip.loss_forward(pred, gt, ranges, torch.ones(17, device=hip_device), _hip.SN_LOSS_WMSE)
    ranges = ranges[:10].contiguous()
    with pytest.raises(sna.HipLibraryError, match="terms"):
        _hip.loss_forward(pred, gt, ranges, torch.ones(10, device=hip_device), 0)
    with pytest.raises(sna.HipLibraryError, match="HIP device"):
        _hip.loss_forward(pred.cpu(), gt.cpu(), ranges, torch.ones(10, device=hip_device), 1)


# ------------------------------------------------------------------ the criterion branches of VERDICT r1 #8
EXTRA = np.load(os.path.join(os.path.dirname(__file__), "golden", "geneo_loss_extra.npz"))


@pytest.mark.parametrize("red", ["mean", "sum", "none"])
@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("case", ["b3_f32", "b3_f64", "b1_f64"])
def test_binary_dice_every_power_and_reduction(hip_device, case, p, red):
    """BinaryDiceLoss(p, reduction) against the reference's own values and gradients (dice_loss.py:33-51)."""
    for smooth in (1, 0.5):
        crit = sna.BinaryDiceLoss(smooth=smooth, p=p, reduction=red)
        pred = torch.from_numpy(EXTRA[f"{case}|pred"]).to(hip_device).requires_grad_(True)
        gt = torch.from_numpy(EXTRA[f"{case}|gt"]).to(hip_device)
        loss = crit(pred, gt)
        loss.sum().backward()
        key = f"{case}|dice_p{p}_{red}_s{smooth}"
        ref, gref = EXTRA[key + "|loss"], EXTRA[key + "|grad_pred"]
        tol = _tol(pred.dtype)
        assert loss.shape == ref.shape
        assert np.abs(loss.detach().cpu().numpy() - ref).max() <= tol * np.abs(ref).max()
        assert np.abs(pred.grad.cpu().numpy() - gref).max() <= tol * np.abs(gref).max()
    # binary occupancy targets (torch.bool, what ToFullDense yields on the device) give the same numbers
    crit = sna.BinaryDiceLoss(p=p, reduction=red)
    pred = torch.from_numpy(EXTRA[f"{case}|pred"]).to(hip_device)
    gt = torch.from_numpy(EXTRA[f"{case}|gt"]).to(hip_device)
    a, b = crit(pred, gt), crit(pred, gt.bool())
    assert torch.allclose(a, b, rtol=1e-6, atol=0)
    with pytest.raises(Exception, match="Unexpected reduction"):
        sna.BinaryDiceLoss(reduction="median")(pred, gt)


@pytest.mark.parametrize("case", ["b1_f32", "b1_f64"])
def test_dice_bce_reduction_none(hip_device, case):
    """BinaryDiceLoss_BCE(reduction='none') and GENEO_Dice_BCE in all three reductions."""
    gt = torch.from_numpy(EXTRA[f"{case}|gt"])
    freqs = torch.from_numpy(EXTRA[f"{case}|dice_bce_none|freqs"])
    crit = sna.BinaryDiceLoss_BCE(targets=gt, weighting_scheme_path=None, save_weighting_scheme=False, reduction="none",
                                  weight_alpha=0.5, weight_epsilon=0.2)
    crit.freqs = freqs.to(hip_device)
    pred = torch.from_numpy(EXTRA[f"{case}|pred"]).to(hip_device).requires_grad_(True)
    loss = crit(pred, gt.to(hip_device))
    loss.sum().backward()
    ref, gref = EXTRA[f"{case}|dice_bce_none|loss"], EXTRA[f"{case}|dice_bce_none|grad_pred"]
    tol = _tol(pred.dtype)
    assert loss.shape == ref.shape
    assert np.abs(loss.detach().cpu().numpy() - ref).max() <= tol * np.abs(ref).max()
    assert np.abs(pred.grad.cpu().numpy() - gref).max() <= tol * np.abs(gref).max()
    # GENEO_Dice_BCE = mse_weight * BinaryDiceLoss_BCE + penalties, in every reduction
    cvx = torch.nn.ParameterDict({"lambda_a": torch.nn.Parameter(torch.tensor(-0.2)),
                                  "lambda_b": torch.nn.Parameter(torch.tensor(1.2), requires_grad=False)}).to(hip_device)
    gp = torch.nn.ParameterDict({"r": torch.nn.Parameter(torch.tensor(-0.5))}).to(hip_device)
    for red in ("mean", "sum", "none"):
        g = sna.GENEO_Dice_BCE(targets=gt, weighting_scheme_path=None, save_weighting_scheme=False, reduction=red,
                               weight_alpha=0.5, weight_epsilon=0.2, mse_weight=3.0, convex_weight=2.0)
        g.freqs = freqs.to(hip_device)
        d = sna.BinaryDiceLoss_BCE(targets=gt, weighting_scheme_path=None, save_weighting_scheme=False, reduction=red,
                                   weight_alpha=0.5, weight_epsilon=0.2)
        d.freqs = freqs.to(hip_device)
        p2 = pred.detach()
        want = 3.0 * d(p2, gt.to(hip_device)) + 2.0 * (0.2 + 0.0) + 2.0 * 0.5   # relu(-lambda_a), relu(-(1 - lambda_a)), relu(-r)
        got = g(p2, gt.to(hip_device), cvx, gp)
        assert got.shape == want.shape
        assert torch.allclose(got, want, rtol=1e-6, atol=0), red
