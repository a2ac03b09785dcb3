"""Host-side mirror of the reference's training criteria for SCENE-Net, over csrc/loss.hip.

Same class names, constructor arguments, `forward` signatures and hyper-parameter meaning as
core/criterions/{w_mse,tversky_loss,dice_loss,geneo_loss}.py, so `LitSceneNet` (lit_model_wrappers.py:160-200) takes
them unchanged.  What differs is how the numbers are produced: every dense term (weighted MSE, focal Tversky, dice) of
one criterion comes out of ONE streaming pass over (pred, gt) on the GPU (`sn_loss_forward`), and the gradient w.r.t.
the prediction out of a second one (`sn_loss_backward`); sums are fp64 in a fixed order.  The penalties over the ~50
scalar parameters (cvx_loss, positive_regularizer) are three torch ops on a stacked vector.

There is no CPU path: pred / gt must be HIP tensors.
"""
from __future__ import annotations

import os
import pickle
from typing import Optional, Tuple

import torch

from . import _hip

HIST_PATH = os.path.join(os.getcwd(), "hist_estimation.pickle")  # w_mse.py:21
ALPHA, BETA, GAMMA = 0.5, 1, 2  # tversky_loss.py:5-7


def save_pickle(data, filename):
    with open(filename, "wb") as handle:
        pickle.dump(data, handle)


def load_pickle(filename):
    with open(filename, "rb") as handle:
        return pickle.load(handle)


class _DenseLossFn(torch.autograd.Function):
    """loss [4] = {sum of terms, wmse, focal tversky, dice} as one differentiable op (gradient flows from any of the
    four entries; they are linear in each other's coefficients only through entry 0, which is what criteria use)."""

    @staticmethod
    def forward(ctx, pred, gt, ranges, bin_w, terms, cfg):
        pred_c, gt_c = pred.contiguous(), gt.contiguous()
        loss, stats, coef, loss32 = _hip.loss_forward(pred_c, gt_c, ranges, bin_w, terms, **cfg)
        ctx.save_for_backward(pred_c, gt_c, ranges, coef)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)   # (no zero-filled gradient for `stats`: a launch per step for nothing)
        # the loss scalar: pred's float dtype; fp32 for bf16 predictions (bf16 is activation storage only) -- the kernel
        # writes both roundings, the cast is not a launch of its own
        return (loss[0] if pred.dtype == torch.float64 else loss32[0]), stats

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        if g_loss is None:
            return None, None, None, None, None, None
        pred, gt, ranges, coef = ctx.saved_tensors
        up = g_loss.detach().reshape(1)   # f64 or f32: read as it is (sn_loss_backward_u)
        return _hip.loss_backward(pred, gt, ranges, coef, up), None, None, None, None, None


def _dense_loss(pred: torch.Tensor, gt: torch.Tensor, ranges: torch.Tensor, bin_w: torch.Tensor, terms: int, **cfg):
    pred, gt = torch.broadcast_tensors(pred, gt)  # w_mse.py:147
    if pred.dim() == 0:
        pred, gt = pred.reshape(1, 1), gt.reshape(1, 1)
    elif pred.dim() == 1:
        pred, gt = pred.reshape(1, -1), gt.reshape(1, -1)
    loss, _ = _DenseLossFn.apply(pred, gt, ranges, bin_w, terms, cfg)
    return loss


class _CriterionFn(torch.autograd.Function):
    """dense terms + penalties over the packed parameter vector as ONE differentiable op in the dense loss's launches
    (sn_criterion_forward / sn_criterion_backward): the same numbers, bit for bit, as _DenseLossFn + _PenaltyFn + the float32
    add between them and the multiply in the penalty's backward -- which a replayed training step paid ~4 us of GPU time each
    for.  float32 / bf16 predictions (the total is a float32 scalar)."""

    @staticmethod
    def forward(ctx, pred, gt, P, ranges, bin_w, terms, cfg, mask, weight, with_sum):
        pred_c, gt_c = pred.contiguous(), gt.contiguous()
        total, stats, coef, pen_grad = _hip.criterion_forward(pred_c, gt_c, ranges, bin_w, terms, P.detach().contiguous(),
                                                              mask, weight, with_sum, **cfg)
        ctx.save_for_backward(pred_c, gt_c, ranges, coef, pen_grad)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return total.reshape(()), stats

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        if g_loss is None:
            return (None,) * 10
        pred, gt, ranges, coef, pen_grad = ctx.saved_tensors
        dpred, dP = _hip.criterion_backward(pred, gt, ranges, coef, g_loss.detach().reshape(1), pen_grad)
        return dpred, None, dP, None, None, None, None, None, None, None


class _PenaltyFn(torch.autograd.Function):
    """geneo_loss.py:36-70 over the packed parameter vector (sn_param_penalty): value and gradient in one launch."""

    @staticmethod
    def forward(ctx, P, mask, weight, with_sum):
        value, grad = _hip.param_penalty(P.detach().contiguous(), mask, weight, with_sum)
        ctx.save_for_backward(grad)
        return value.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


def _live_pack(container):
    """(P, live) when `container` came from a scene_net_amd model whose latest differentiable forward gathered its
    parameters into one vector P that is still current (scene_net._LivePack), else (None, None)."""
    live = getattr(container, "_sn_live", None)
    P = live.current() if live is not None else None
    return (P, live) if P is not None else (None, None)


_UNIT = {}


def _unit_tables(device) -> Tuple[torch.Tensor, torch.Tensor]:
    """A one-bin table for criteria that carry no weighting scheme (Tversky / dice alone)."""
    key = str(device)
    if key not in _UNIT:
        _UNIT[key] = (torch.zeros(1, dtype=torch.float32, device=device), torch.ones(1, dtype=torch.float32, device=device))
    return _UNIT[key]


class WeightedMSE(torch.nn.Module):
    """w_mse.py:24-151.  Weighted MSE whose weights come from an inverse-frequency histogram of the targets."""

    def __init__(self, targets=None, weighting_scheme_path=HIST_PATH, weight_alpha=1, weight_epsilon=0.1, mse_weight=1,
                 **kwargs) -> None:
        super().__init__()
        self.weight_alpha = weight_alpha
        self.weight_epsilon = weight_epsilon
        self.mse_weight = mse_weight
        self.relu = torch.nn.ReLU()
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")  # w_mse.py:57
        if weighting_scheme_path is not None and os.path.exists(weighting_scheme_path):
            self.pik_name = weighting_scheme_path
            self.freqs, self.ranges = load_pickle(self.pik_name)
        elif targets is not None:
            self.freqs, self.ranges = self.hist_frequency_estimation(torch.flatten(targets), plot=False)
            if kwargs.get("save_weighting_scheme", True):  # w_mse.py:65 writes ./hist_estimation.pickle
                save_pickle((self.freqs.cpu(), self.ranges.cpu()), os.path.join(".", "hist_estimation.pickle"))
        else:
            # the reference builds this ValueError without raising it (w_mse.py:67) and dies on the next line
            raise ValueError("No targets were provided to build the weighting scheme")
        self.freqs = self.freqs.to(self.device)
        self.ranges = self.ranges.to(self.device)
        self._tables = {}

    def hist_frequency_estimation(self, y: torch.Tensor, hist_len=10, plot=False):
        """w_mse.py:72-112: counts per bin int(hist_len*y); ranges = left edges."""
        hist_range = torch.linspace(0, 1, hist_len + 1, device=self.device)[:-1]
        y = y.to(self.device)
        hist_idxs = (hist_len * y).to(torch.int)
        hist_count = torch.bincount(hist_idxs, minlength=hist_len)
        if plot:
            print("Histogram Bin /\t Count")
            step = hist_range[1] - hist_range[0]
            for i in range(len(hist_range)):
                print(f"[{hist_range[i]:.3f}, {hist_range[i] + step:.3f}[ : {hist_count[i]}")
        return hist_count, hist_range

    # -- the ~10-entry tables everything else derives from ------------------------------------------------------
    def _bin_values(self) -> torch.Tensor:
        """Value each starting bin ends with after w_mse.py:124-126, whose in-place loop re-replaces a frequency
        that equals a later bin index."""
        freqs = [int(f) for f in self.freqs.tolist()]
        out = []
        for k in range(len(self.ranges)):
            v = k
            for idx, f in enumerate(freqs):
                if v == idx:
                    v = f
            out.append(v)
        return torch.tensor(out, dtype=torch.int64)

    def _bin_weights(self) -> torch.Tensor:
        """max(1 - alpha*dens, eps) per bin in fp32 (w_mse.py:128-142), before the division by the mean."""
        vals = self._bin_values()
        freqs = self.freqs.cpu()
        fmin, fmax = torch.min(freqs), torch.max(freqs)
        dens = (vals - fmin) / (fmax - fmin)
        return torch.max(1 - self.weight_alpha * dens, torch.full_like(dens, self.weight_epsilon))

    def _device_tables(self, device) -> Tuple[torch.Tensor, torch.Tensor]:
        key = (str(device), float(self.weight_alpha), float(self.weight_epsilon), id(self.freqs), self.freqs._version,
               id(self.ranges))
        if key not in self._tables:
            self._tables.clear()
            self._tables[key] = (self.ranges.to(device=device, dtype=torch.float32).contiguous(),
                                 self._bin_weights().to(device=device, dtype=torch.float32).contiguous())
        return self._tables[key]

    def get_dens_target(self, y: torch.Tensor, calc_weights=False):
        """w_mse.py:114-131 (table lookup instead of the in-place loop; same values)."""
        if calc_weights:
            self.freqs, self.ranges = self.hist_frequency_estimation(y)
            self._tables.clear()
        hist_idx = torch.abs(torch.unsqueeze(y, -1) - self.ranges.to(y.device)).argmin(dim=-1)
        vals = self._bin_values().to(y.device)[hist_idx]
        freq_min, freq_max = torch.min(self.freqs), torch.max(self.freqs)
        return (vals - freq_min.to(y.device)) / (freq_max - freq_min).to(y.device)

    def get_weight_target(self, y: torch.Tensor):
        """w_mse.py:133-144."""
        y = y.to(self.device)
        y_dens = self.get_dens_target(y)
        weights = torch.max(1 - self.weight_alpha * y_dens, torch.full_like(y_dens, self.weight_epsilon))
        return weights / torch.mean(weights)

    def _dense(self, y_pred, y_gt, terms, **cfg):
        ranges, bin_w = self._device_tables(y_pred.device)
        return _dense_loss(y_pred, y_gt, ranges, bin_w, terms, mse_weight=self.mse_weight, **cfg)

    def forward(self, y_pred: torch.Tensor, y_gt: torch.Tensor):
        """w_mse.py:146-151."""
        return self._dense(y_pred, y_gt, _hip.SN_LOSS_WMSE)

    @staticmethod
    def add_model_specific_args(parent_parser):
        parser = parent_parser.add_argument_group("WeightedMSE")
        parser.add_argument("--weight_alpha", type=float, default=1)
        parser.add_argument("--weight_epsilon", type=float, default=0.01)
        parser.add_argument("--mse_weight", type=float, default=1)
        parser.add_argument("--hist_path", type=str, default=HIST_PATH)
        return parent_parser


class TverskyLoss(torch.nn.Module):
    """tversky_loss.py:10-60."""

    def __init__(self, tversky_alpha=ALPHA, tversky_beta=BETA, tversky_smooth=1, **kwargs):
        super().__init__()
        self.tversky_alpha = tversky_alpha
        self.tversky_beta = tversky_beta
        self.tversky_smooth = tversky_smooth

    def _cfg(self, gamma=1.0):
        return dict(tversky_alpha=self.tversky_alpha, tversky_beta=self.tversky_beta, focal_gamma=gamma,
                    tversky_smooth=self.tversky_smooth)

    def forward(self, inputs, targets):
        ranges, bin_w = _unit_tables(inputs.device)
        return _dense_loss(inputs, targets, ranges, bin_w, _hip.SN_LOSS_FOCAL_TVERSKY, **self._cfg(1.0))

    @staticmethod
    def add_model_specific_args(parent_parser):
        parser = parent_parser.add_argument_group("TverskyLoss")
        parser.add_argument("--tversky_alpha", type=float, default=ALPHA)
        parser.add_argument("--tversky_beta", type=float, default=BETA)
        parser.add_argument("--tversky_smooth", type=float, default=1)
        return parent_parser


class FocalTverskyLoss(TverskyLoss):
    """tversky_loss.py:63-103."""

    def __init__(self, tversky_alpha=ALPHA, tversky_beta=BETA, focal_gamma=GAMMA, tversky_smooth=1, **kwargs):
        super().__init__(tversky_alpha, tversky_beta, tversky_smooth)
        self.focal_gamma = focal_gamma

    def forward(self, inputs, targets):
        ranges, bin_w = _unit_tables(inputs.device)
        return _dense_loss(inputs, targets, ranges, bin_w, _hip.SN_LOSS_FOCAL_TVERSKY, **self._cfg(self.focal_gamma))

    @staticmethod
    def add_model_specific_args(parent_parser):
        parser = parent_parser.add_argument_group("FocalTverskyLoss")
        parser.add_argument("--tversky_alpha", type=float, default=ALPHA)
        parser.add_argument("--tversky_beta", type=float, default=BETA)
        parser.add_argument("--focal_gamma", type=float, default=GAMMA)
        parser.add_argument("--tversky_smooth", type=float, default=1)
        return parent_parser


def _dice_per_sample(predict: torch.Tensor, target: torch.Tensor, smooth, power) -> torch.Tensor:
    """dice_loss.py:35-41: 1 - (sum(p t) + s) / (sum(p^power + t^power) + s) per sample, as torch ops on the device
    (the branches the fused kernel does not serve: power != 2, reduction 'none')."""
    predict = predict.contiguous().view(predict.shape[0], -1)
    target = target.contiguous().view(target.shape[0], -1)
    if not target.dtype.is_floating_point:
        target = target.to(predict.dtype)
    num = torch.sum(torch.mul(predict, target), dim=1) + smooth
    den = torch.sum(predict.pow(power) + target.pow(power), dim=1) + smooth
    return 1 - num / den


class BinaryDiceLoss(torch.nn.Module):
    """dice_loss.py:8-51.  p = 2 with reduction 'mean' / 'sum' (what every criterion of resolve_criterion uses) is one
    streaming pass of csrc/loss.hip; any other power and reduction 'none' are the reference's formula in torch ops on
    the device.  An unknown reduction raises at call time, like the reference."""

    def __init__(self, smooth=1, p=2, reduction="mean", **kwargs):
        super().__init__()
        self.smooth = smooth
        self.power = p
        self.reduction = reduction

    def forward(self, predict, target):
        assert predict.shape[0] == target.shape[0], "predict & target batch size don't match"
        if self.reduction not in ("mean", "sum", "none"):
            raise Exception("Unexpected reduction {}".format(self.reduction))
        if not predict.is_cuda:
            raise _hip.HipLibraryError("BinaryDiceLoss runs on the HIP device only (no CPU path): move predict to cuda")
        if self.power == 2 and self.reduction != "none":
            ranges, bin_w = _unit_tables(predict.device)
            loss = _dense_loss(predict, target, ranges, bin_w, _hip.SN_LOSS_DICE, dice_smooth=self.smooth)
            return loss if self.reduction == "mean" else loss * predict.shape[0]
        loss = _dice_per_sample(predict, target, self.smooth, self.power)
        if self.reduction == "mean":
            return loss.mean()
        return loss.sum() if self.reduction == "sum" else loss


class BinaryDiceLoss_BCE(WeightedMSE):
    """dice_loss.py:55-95: weights * BCELoss + BinaryDiceLoss.  'mean' / 'sum': one pass over (pred, gt) in
    csrc/loss.hip; 'none': the reference's expression in torch ops (per-element weighted BCE plus the per-sample dice
    vector, with whatever torch's broadcasting makes of that sum)."""

    def __init__(self, targets=None, weighting_scheme_path=HIST_PATH, weight_alpha=1, weight_epsilon=0.1, mse_weight=1,
                 reduction="mean", **kwargs) -> None:
        super().__init__(targets, weighting_scheme_path, weight_alpha, weight_epsilon, mse_weight, **kwargs)
        self.dice = BinaryDiceLoss(reduction=reduction)
        self.reduction = reduction

    def forward(self, predict, target):
        if self.reduction not in ("mean", "sum", "none"):
            raise Exception("Unexpected reduction {}".format(self.reduction))
        if self.reduction == "none":
            tgt = target if target.dtype.is_floating_point else target.to(predict.dtype)
            weights = self.get_weight_target(tgt).to(predict.device)
            bce = torch.nn.functional.binary_cross_entropy(predict, tgt.to(predict.dtype), reduction="none")
            return weights * bce + self.dice(predict, target)
        ranges, bin_w = self._device_tables(predict.device)
        if self.reduction == "mean":   # mean(w * bce) + mean_b(dice_b)
            return _dense_loss(predict, target, ranges, bin_w, _hip.SN_LOSS_WBCE | _hip.SN_LOSS_DICE,
                               dice_smooth=self.dice.smooth)
        # 'sum': sum(w * bce) + sum_b(dice_b) = n * mean(w * bce) + B * mean_b(dice_b)
        bce = _dense_loss(predict, target, ranges, bin_w, _hip.SN_LOSS_WBCE)
        return bce * predict.numel() + self.dice(predict, target)


class GENEO_Loss(WeightedMSE):
    """geneo_loss.py:24-90: weighted MSE + penalties on non-positive convex coefficients / GENEO parameters."""

    def __init__(self, targets=None, weighting_scheme_path=HIST_PATH, weight_alpha=1, weight_epsilon=0.1, mse_weight=1,
                 convex_weight=1, **kwargs) -> None:
        super().__init__(targets, weighting_scheme_path, weight_alpha, weight_epsilon, mse_weight, **kwargs)
        self.cvx_w = convex_weight

    def cvx_loss(self, cvx_coeffs):
        """geneo_loss.py:36-61: relu(-phi) over the trainable coefficients, plus relu(-(1 - sum of them)) for the
        frozen last one."""
        if len(cvx_coeffs) == 0:
            return 0
        P, live = _live_pack(cvx_coeffs)
        if P is not None:   # the model's gathered parameter vector: one launch, one autograd node
            return _PenaltyFn.apply(P, live.mask_cvx, self.cvx_w, True)
        last_phi = [n for n in cvx_coeffs if not cvx_coeffs[n].requires_grad][0]
        free = torch.stack([phi for n, phi in cvx_coeffs.items() if n != last_phi])
        return self.cvx_w * (torch.relu(-free).sum() + torch.relu(-(1 - free.sum())))

    def positive_regularizer(self, params):
        """geneo_loss.py:63-70."""
        if len(params) == 0:
            return 0
        P, live = _live_pack(params)
        if P is not None:
            return _PenaltyFn.apply(P, live.mask_params, self.cvx_w, False)
        return self.cvx_w * torch.relu(-torch.stack(list(params.values()))).sum()

    def _penalties(self, cvx_coeffs, geneo_params):
        """cvx_loss + positive_regularizer; one launch when both come from the same live scene_net_amd model."""
        Pc, live = _live_pack(cvx_coeffs)
        Pp, live_p = _live_pack(geneo_params)
        if Pc is not None and Pp is Pc and len(cvx_coeffs) and len(geneo_params):
            return _PenaltyFn.apply(Pc, live.mask_all, self.cvx_w, True)
        return self.cvx_loss(cvx_coeffs) + self.positive_regularizer(geneo_params)

    def _terms(self):
        return _hip.SN_LOSS_WMSE, {}

    def forward(self, y_pred, y_gt, cvx_coeffs, geneo_params):
        terms, cfg = self._terms()
        Pc, live = _live_pack(cvx_coeffs)
        Pp, _ = _live_pack(geneo_params)
        if (Pc is not None and Pp is Pc and len(cvx_coeffs) and len(geneo_params) and y_pred.is_cuda
                and y_pred.dtype in (torch.float32, torch.bfloat16) and y_pred.dim() >= 2 and y_pred.shape == y_gt.shape):
            # one differentiable op in the dense loss's own launches (the common training step)
            ranges, bin_w = self._device_tables(y_pred.device)
            loss, _ = _CriterionFn.apply(y_pred, y_gt, Pc, ranges, bin_w, terms, dict(mse_weight=self.mse_weight, **cfg),
                                         live.mask_all, self.cvx_w, True)
            return loss
        dense_criterion = self._dense(y_pred, y_gt, terms, **cfg)
        return dense_criterion + self._penalties(cvx_coeffs, geneo_params)

    def __str__(self):
        return f"GENEO Loss with mse_weight={self.mse_weight} and alpha={self.weight_alpha} and epsilon={self.weight_epsilon}"

    @staticmethod
    def add_model_specific_args(parent_parser):
        parent_parser = WeightedMSE.add_model_specific_args(parent_parser)
        parser = parent_parser.add_argument_group("GENEO_Loss")
        parser.add_argument("--cvx_w", type=float, default=1.0)
        return parent_parser


class GENEO_Dice_Loss(GENEO_Loss):
    """geneo_loss.py:131-143: weighted MSE + BinaryDiceLoss + penalties -- one pass over (pred, gt)."""

    def __init__(self, targets=None, weighting_scheme_path=None, weight_alpha=1, weight_epsilon=0.1, mse_weight=1,
                 convex_weight=1, **kwargs) -> None:
        super().__init__(targets, weighting_scheme_path, weight_alpha, weight_epsilon, mse_weight, convex_weight,
                         **kwargs)
        self.dice = BinaryDiceLoss()

    def _terms(self):
        return _hip.SN_LOSS_WMSE | _hip.SN_LOSS_DICE, dict(dice_smooth=self.dice.smooth)


class GENEO_Dice_BCE(GENEO_Loss):
    """geneo_loss.py:110-128: mse_weight * (weights * BCE + dice) + penalties, reduction 'mean' / 'sum' / 'none' as in
    BinaryDiceLoss_BCE.  (The reference's constructor passes its arguments to BinaryDiceLoss_BCE in the wrong
    positions and raises a TypeError; this is the evident intent.)"""

    def __init__(self, targets=None, weighting_scheme_path=None, weight_alpha=1, weight_epsilon=0.1, mse_weight=1,
                 convex_weight=1, reduction="mean", **kwargs) -> None:
        super().__init__(targets, weighting_scheme_path, weight_alpha, weight_epsilon, mse_weight, convex_weight,
                         **kwargs)
        self.reduction = reduction
        self.dice = BinaryDiceLoss(reduction=reduction)

    def forward(self, y_pred, y_gt, cvx_coeffs, geneo_params):
        # the same three reductions as BinaryDiceLoss_BCE.forward, on this object's own weighting tables
        dense = BinaryDiceLoss_BCE.forward(self, y_pred, y_gt)
        return self.mse_weight * dense + self._penalties(cvx_coeffs, geneo_params)


class GENEO_Tversky_Loss(GENEO_Loss):
    """geneo_loss.py:145-161: weighted MSE + FocalTverskyLoss + penalties -- one pass over (pred, gt)."""

    def __init__(self, targets=None, weighting_scheme_path=None, weight_alpha=1, weight_epsilon=0.1, mse_weight=1,
                 convex_weight=1, tversky_alpha=0.5, tversky_beta=1, focal_gamma=1, tversky_smooth=1, **kwargs) -> None:
        super().__init__(targets, weighting_scheme_path, weight_alpha, weight_epsilon, mse_weight, convex_weight,
                         **kwargs)
        self.tversky = FocalTverskyLoss(tversky_alpha, tversky_beta, focal_gamma, tversky_smooth)

    def _terms(self):
        return _hip.SN_LOSS_WMSE | _hip.SN_LOSS_FOCAL_TVERSKY, self.tversky._cfg(self.tversky.focal_gamma)

    @staticmethod
    def add_model_specific_args(parent_parser):
        parent_parser = GENEO_Loss.add_model_specific_args(parent_parser)
        return FocalTverskyLoss.add_model_specific_args(parent_parser)
