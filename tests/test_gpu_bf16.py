"""bf16 activation storage in the training path (BASELINE C5; reference: `precision: 16`,
experiments/scenenet_ts40k/defaults_config.yml:83-84; criterion core/criterions/geneo_loss.py:145-161).
The prediction, its gradient and what the backward correlation reads are bf16; every sum stays fp32 / fp64.
Stated tolerance: a bf16 value carries 8 significant bits, so a stored prediction is within 2^-9 relative (2e-3
absolute on [0, 1)) of the fp32 one; sums over the grid average that rounding out."""
import numpy as np
import pytest
import torch

import scene_net_amd as sna
from scene_net_amd import _hip
from oracle import geneo_oracle as go
from oracle import loss_oracle as lo

pytestmark = pytest.mark.gpu
BF16_ABS = 2.0 ** -8     # |bf16(v) - v| <= 2^-9 |v|; predictions are in [0, 1)


def _bank(dev):
    from scene_net_amd.synthetic import synthetic_bank_spec
    specs, names, lambdas, last = synthetic_bank_spec()
    bank = go.geneo_bank(specs, (9, 9, 9))[:, 0].float()
    return bank, go.effective_lambdas(lambdas, last, names)


def test_linear_forward_writes_bf16_within_one_rounding(hip_device):
    torch.manual_seed(0)
    bank, lam = _bank(hip_device)
    occ = torch.rand(2, 1, 24, 24, 64) < 0.06
    ref_act = go.conv_bank(occ.double(), bank.double().unsqueeze(1))
    ref = torch.relu(torch.tanh((lam.double().view(1, -1, 1, 1, 1) * ref_act).sum(1, keepdim=True)))
    x, b, l = occ.to(hip_device), bank.to(hip_device).contiguous(), lam.float().to(hip_device)
    o32 = _hip.conv_fused(x, b, l)
    o16 = _hip.conv_fused(x, b, l, out_dtype=torch.bfloat16)
    assert o16.dtype == torch.bfloat16 and o16.shape == o32.shape
    assert torch.equal(o16, o32.to(torch.bfloat16))          # the same fp32 value, rounded once (round to nearest even)
    assert (o16.double().cpu() - ref).abs().max().item() <= BF16_ABS / 2 + 1e-4


@pytest.mark.parametrize("gt_kind", ["bool", "f32"])
def test_criterion_on_bf16_predictions(hip_device, gt_kind):
    """sn_loss_forward / sn_loss_backward with bf16 pred: the loss equals the fp64 oracle's on the SAME (bf16-rounded)
    predictions, the gradient is the fp32 gradient rounded to bf16."""
    torch.manual_seed(1)
    shape = (3, 1, 8, 16, 32)
    pred16 = torch.rand(shape).clamp(1e-3, 1 - 1e-3).to(torch.bfloat16)
    gtb = torch.rand(shape) < 0.1
    gt = gtb if gt_kind == "bool" else torch.where(torch.rand(shape) < 0.5, gtb.float(), torch.rand(shape) * gtb.float())
    crit = sna.GENEO_Tversky_Loss(targets=gt.float(), weighting_scheme_path=None, save_weighting_scheme=False,
                                  weight_alpha=0.7, focal_gamma=1.5)
    cvx, gp = {}, {}
    p16 = pred16.to(hip_device).requires_grad_(True)
    loss16 = crit(p16, gt.to(hip_device), cvx, gp)
    loss16.backward()
    assert loss16.dtype == torch.float32 and p16.grad.dtype == torch.bfloat16
    p32 = pred16.float().to(hip_device).requires_grad_(True)
    loss32 = crit(p32, gt.to(hip_device), cvx, gp)
    loss32.backward()
    assert abs(loss16.item() - loss32.item()) <= 1e-6 * abs(loss32.item())       # same values in, fp64 sums
    assert torch.equal(p16.grad, p32.grad.to(torch.bfloat16))
    # and against the oracle in fp64
    po = pred16.double().requires_grad_(True)
    ref = lo.geneo_tversky_loss(po, gt.double(), cvx, gp, crit.freqs.cpu(), crit.ranges.cpu(), 0.7, 0.1, 1.0, 1.0,
                                0.5, 1.0, 1.5, 1.0)
    assert abs(loss16.item() - float(ref)) <= 2e-5 * abs(float(ref))


def test_backward_correlation_reads_bf16(hip_device):
    """sn_conv_corr_t on bf16 (gout, out): bit-identical to the same values widened to fp32 (products and sums are fp32
    in both cases)."""
    torch.manual_seed(2)
    x = (torch.rand(2, 1, 16, 16, 64) < 0.05).to(hip_device)
    gout = (torch.randn(2, 1, 16, 16, 64) * 1e-3).to(torch.bfloat16).to(hip_device)
    out = torch.rand(2, 1, 16, 16, 64).to(torch.bfloat16).to(hip_device)
    c16 = _hip.conv_corr(x, gout, out, (9, 9, 9))
    c32 = _hip.conv_corr(x, gout.float(), out.float(), (9, 9, 9))
    assert torch.equal(c16, c32)
    with pytest.raises(sna.HipLibraryError):
        _hip.conv_corr(x, gout, out.float(), (9, 9, 9))     # mixed dtypes are refused


def _abs_sum_bounds(model, x, gout, out, dev, ks=(9, 9, 9)):
    """B_p per trainable scalar (see test_training_step_with_bf16_activations): the correlation of |delta| through |lambda_g|
    and the absolute Jacobians of the bank (central differences on sn_geneo_bank); the coefficients' own gradient is
    <K_g, C> - <K_last, C> (the frozen coefficient is 1 - sum of the others, SCENE_Net.py:331)."""
    C_abs = _hip.conv_corr(x, gout.float().abs().contiguous(), out.float().contiguous(), ks)
    lam = model.effective_lambdas(dev).clone()
    names = list(model.geneos)
    last = names.index(model.last_lambda.replace("lambda_", "", 1))
    bank0 = model.compute_bank(dev).clone()
    B = {}
    with torch.no_grad():
        for g, gname in enumerate(names):
            for pname, p in model.geneos[gname].geneo_params.items():
                if not p.requires_grad:
                    continue
                v0 = float(p)
                h = 1e-3 * max(1.0, abs(v0))
                p.fill_(v0 + h)
                kp = model.compute_bank(dev)[g].clone()
                p.fill_(v0 - h)
                km = model.compute_bank(dev)[g].clone()
                p.fill_(v0)
                B[f"geneos.{gname}.geneo_params.{pname}"] = float((lam[g].abs() * ((kp - km) / (2 * h)).abs() * C_abs).sum())
        for g, gname in enumerate(names):
            if g != last:
                B[f"lambdas_dict.lambda_{gname}"] = float(((bank0[g].abs() + bank0[last].abs()) * C_abs).sum())
    return B


def test_training_step_with_bf16_activations(hip_device):
    """One full step (voxelise + GT, forward, GENEO_Tversky_Loss, backward) with SceneNet.activation_dtype = bfloat16
    against the same step in fp32, every trainable scalar's gradient held to the bound ONE bf16 rounding of the prediction
    and of its gradient predicts for THAT scalar:

        g_p = sum_g sum_tap lambda_g dK_g/dp[tap] C[tap],   C[tap] = sum_v delta_v x[v + tap]   (x binary),
        delta_v = dL/dpred_v (1 - out_v^2) [out_v > 0];   bf16 storage moves every delta_v by <= ~2 x 2^-9 |delta_v|, so
        |g16_p - g32_p| <= eps B_p,   B_p = sum_g sum_tap |lambda_g dK_g/dp[tap]| C_abs[tap],  C_abs = the correlation of |delta|

    -- B_p has no cancellation left, which is what made the old bar (2 % of the NET gradient, up to 5 % measured where the
    sum nearly cancels) so loose.  eps: worst case 2 x 2^-9; the roundings are independent, and [measured, three seeds,
    tools/debug/bf16_grad_bound.py] the step lands at <= 0.12 x 2^-9 of B_p for every scalar.  The bar is 0.5 x 2^-9.
    What pins the roundings themselves are the exact tests above (bf16 output == fp32 output rounded once, loss gradient ==
    fp32 gradient rounded once, sn_conv_corr_t on bf16 == the same values widened)."""
    from scene_net_amd.synthetic import synthetic_tile
    tiles, labels = zip(*[synthetic_tile(40 + i, 20_000) for i in range(4)])
    batch = sna.PointBatch.from_tiles(tiles, labels, device=hip_device)
    grads = {}
    for dt in (None, torch.bfloat16):
        torch.manual_seed(5)
        model = sna.SceneNet({"cy": 2, "cone": 2, "neg": 1}, (9, 9, 9)).to(hip_device)
        model.activation_dtype = dt
        pipe = sna.ScenePipeline(model, (32, 32, 64), keep_labels=[15.0])
        grids = pipe.voxelize(batch, want_gt=True)
        crit = sna.GENEO_Tversky_Loss(targets=torch.tensor([0.0, 1.0]), weighting_scheme_path=None,
                                      save_weighting_scheme=False)
        pred = model(grids.occ)
        assert pred.dtype == (torch.bfloat16 if dt is not None else torch.float32)
        pred.retain_grad()
        loss = crit(pred, grids.gt_occ, model.get_cvx_coefficients(), model.get_geneo_params())
        loss.backward()
        grads[dt] = ({n: p.grad.item() for n, p in model.named_parameters() if p.grad is not None}, loss.item())
        if dt is None:
            bounds = _abs_sum_bounds(model, grids.occ, pred.grad.detach(), pred.detach(), hip_device)
    (g32, l32), (g16, l16) = grads[None], grads[torch.bfloat16]
    assert abs(l16 - l32) <= 2e-3 * abs(l32)
    assert set(g16) == set(g32) and len(g32) >= 12 and set(bounds) >= set(g32)
    eps = 0.5 * 2.0 ** -9
    for n in g32:
        assert abs(g16[n] - g32[n]) <= eps * bounds[n] + 1e-7, (n, g16[n], g32[n], bounds[n])
    # the bound is a real constraint: for most scalars it is far below the old 2 % of the net gradient
    tighter = sum(1 for n in g32 if eps * bounds[n] < 2e-2 * abs(g32[n]))
    assert tighter >= len(g32) // 2, (tighter, len(g32))
    # the whole step also captures and replays with bf16 activations
    torch.manual_seed(5)
    model = sna.SceneNet({"cy": 2, "cone": 2, "neg": 1}, (9, 9, 9)).to(hip_device)
    model.activation_dtype = torch.bfloat16
    pipe = sna.ScenePipeline(model, (32, 32, 64), keep_labels=[15.0])
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    step = sna.CapturedTrainingStep(pipe, crit, opt, batch, warmup=2)
    l_a = float(step.replay())
    l_b = float(step.replay())
    assert np.isfinite(l_a) and np.isfinite(l_b)
