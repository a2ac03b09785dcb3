"""
Generates tests/golden/geneo_*.npz by IMPORTING the reference (dlavado/scene-net at
/root/reference) in the build container.  The reference never travels: only the
arrays written here are committed.  Run from the repo root:

    python tests/golden/make_golden.py

Third-party modules the reference imports for plotting / IO only (pyntcloud,
open3d, laspy, ...) are absent from the image and are replaced by MagicMock
before import; the compute path (torch only) is the reference's
own code, executed unmodified.
"""
import importlib
import os
import sys
import warnings
from unittest.mock import MagicMock

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = os.environ.get("SCENENET_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

for name in ["pyntcloud", "open3d", "laspy", "webcolors", "sympytorch", "IPython", "IPython.display", "seaborn",
             "torchvision", "torchvision.transforms", "pytorch_lightning", "pytorch_lightning.callbacks", "wandb",
             "torchmetrics", "torchviz", "torchsummary"]:
    try:
        importlib.import_module(name)
    except Exception:
        sys.modules[name] = MagicMock()

sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

from core.models.geneos import cylinder, arrow, neg_sphere  # noqa: E402
from core.models.SCENE_Net import SceneNet  # noqa: E402

T = lambda v: torch.tensor(v, dtype=torch.float32)  # noqa: E731

KERNEL_SIZES = [(9, 9, 9), (9, 7, 7), (9, 5, 5), (6, 5, 5), (9, 6, 6), (5, 7, 3), (3, 3, 3), (4, 6, 5)]

CY_PARAMS = [dict(radius=2.5, sigma=1.8), dict(radius=0.998896, sigma=1.199054), dict(radius=4.0, sigma=0.5),
             dict(radius=0.5, sigma=2.0)]
CONE_PARAMS = [dict(radius=1.5, sigma=0.955910, apex=0.0, cone_radius=4.000988, cone_inc=0.565547),  # trained ckpt
               dict(radius=2.0, sigma=1.4, apex=4.0, cone_radius=3.0, cone_inc=0.2),
               dict(radius=3.5, sigma=1.0, apex=2.7, cone_radius=1.5, cone_inc=0.45),
               dict(radius=1.0, sigma=2.0, apex=3.0, cone_radius=2.0, cone_inc=0.1),  # smart config, arrow.py:143-151
               dict(radius=2.0, sigma=1.2, apex=1.0, cone_radius=2.5, cone_inc=-0.3)]  # clamp to 0
NEG_PARAMS = [dict(radius=3.000918, sigma=0.605097, neg_factor=0.127053),  # trained ckpt
              dict(radius=2.0, sigma=0.9, neg_factor=0.2), dict(radius=5.0, sigma=0.5, neg_factor=0.9)]


def dump_kernels():
    out = {}
    meta = []
    for ks in KERNEL_SIZES:
        for i, p in enumerate(CY_PARAMS):
            k = cylinder.cylinderv2("cy", ks, **{n: T(v) for n, v in p.items()}).kernel
            key = f"cy|{ks}|{i}"
            out[key] = k.detach().cpu().numpy()
            meta.append((key, "cy", ks, p))
        for i, p in enumerate(CONE_PARAMS):
            if int(p["apex"]) > ks[0]:
                continue
            k = arrow.arrow("cone", ks, **{n: T(v) for n, v in p.items()}).kernel
            key = f"cone|{ks}|{i}"
            out[key] = k.detach().cpu().numpy()
            meta.append((key, "cone", ks, p))
        for i, p in enumerate(NEG_PARAMS):
            k = neg_sphere.negSpherev2("neg", ks, **{n: T(v) for n, v in p.items()}).kernel
            key = f"neg|{ks}|{i}"
            out[key] = k.detach().cpu().numpy()
            meta.append((key, "neg", ks, p))
    # v1 generators (SCENE_Net v1): cylinder_kernel, cone_kernel, neg_sphere_kernel
    for ks in [(9, 9, 9), (9, 5, 5), (6, 5, 6), (4, 6, 5)]:
        for i, p in enumerate(CY_PARAMS):
            k = cylinder.cylinder_kernel("cy", ks, **{n: T(v) for n, v in p.items()}).kernel
            key = f"cy_v1|{ks}|{i}"
            out[key] = k.detach().cpu().numpy(); meta.append((key, "cy_v1", ks, p))
        for i, p in enumerate(CONE_PARAMS):
            if int(p["apex"]) > ks[0] or p["cone_inc"] <= 0:
                continue
            k = arrow.cone_kernel("cone", ks, **{n: T(v) for n, v in p.items()}).kernel
            key = f"cone_v1|{ks}|{i}"
            out[key] = k.detach().cpu().numpy(); meta.append((key, "cone_v1", ks, p))
        for i, p in enumerate(NEG_PARAMS):
            k = neg_sphere.neg_sphere_kernel("neg", ks, **{n: T(v) for n, v in p.items()}).kernel
            key = f"neg_v1|{ks}|{i}"
            out[key] = k.detach().cpu().numpy(); meta.append((key, "neg_v1", ks, p))
    np.savez_compressed(os.path.join(OUT, "geneo_kernels.npz"), **out)
    import json
    with open(os.path.join(OUT, "geneo_kernels_meta.json"), "w") as f:
        json.dump([dict(key=k, kind=kind, kernel_size=list(ks), params=p) for k, kind, ks, p in meta], f, indent=0)
    print(f"geneo_kernels.npz: {len(out)} kernels")


def set_model(model, geneo_params, lambdas, last):
    with torch.no_grad():
        for gname, p in geneo_params.items():
            for pname, v in p.items():
                model.geneos[gname].geneo_params[pname].fill_(v)
        for lname, v in lambdas.items():
            model.lambdas_dict[lname].fill_(v)
    model.last_lambda = last


def forward_case(tag, geneo_num, ks, grid, batch, geneo_params, lambdas, last, seed, density):
    torch.manual_seed(0)
    model = SceneNet(geneo_num, ks)
    set_model(model, geneo_params, lambdas, last)
    rng = np.random.default_rng(seed)
    x = (rng.random((batch, 1) + tuple(grid)) < density).astype(np.float64)
    xt = torch.from_numpy(x)
    with torch.no_grad():
        kernels = torch.stack([model.geneos[g].compute_kernel() for g in model.geneos])
        conv = torch.nn.functional.conv3d(xt, kernels, padding="same")
        out = model(xt)
    names = list(model.geneos.keys())
    return {
        f"{tag}/x": x.astype(np.uint8), f"{tag}/bank": kernels.numpy(), f"{tag}/conv": conv.numpy(),
        f"{tag}/out": out.numpy(), f"{tag}/names": np.array(names),
        f"{tag}/lambdas": np.array([lambdas[f"lambda_{n}"] for n in names], dtype=np.float32),
        f"{tag}/last": np.array(names.index(last.replace("lambda_", ""))),
        f"{tag}/kernel_size": np.array(ks),
        f"{tag}/lambda_last_after": np.array(model.lambdas_dict[last].item(), dtype=np.float32),
    }, {tag: dict(geneo_params=geneo_params, names=names)}


def dump_forward():
    out, meta = {}, {}
    # (1) the trained checkpoint's 13 scalars, kernel (9,5,5), 3 GENEOs
    ckpt_params = {"cy_0": dict(radius=0.998896, sigma=1.199054),
                   "cone_0": dict(apex=0.0, cone_inc=0.565547, cone_radius=4.000988, radius=1.5, sigma=0.955910),
                   "neg_0": dict(neg_factor=0.127053, radius=3.000918, sigma=0.605097)}
    ckpt_lam = {"lambda_cy_0": 0.024178, "lambda_cone_0": 0.608911, "lambda_neg_0": 0.366911}
    a, m = forward_case("ckpt955", {"cy": 1, "cone": 1, "neg": 1}, (9, 5, 5), (24, 20, 16), 2, ckpt_params, ckpt_lam,
                        "lambda_neg_0", seed=11, density=0.08)
    out.update(a); meta.update(m)
    # (2) C1-shaped bank (cy 2, cone 1, neg 1), cubic 9^3, explicit params
    rng = np.random.default_rng(7)
    def rp(kind):
        if kind == "cy":
            return dict(radius=float(rng.uniform(0.5, 4)), sigma=float(rng.uniform(0.5, 2)))
        if kind == "cone":
            return dict(radius=float(rng.uniform(0.5, 4)), sigma=float(rng.uniform(0.5, 2)),
                        apex=float(rng.integers(4, 8)), cone_radius=float(rng.uniform(0.5, 4)),
                        cone_inc=float(rng.uniform(0.05, 0.45)))
        return dict(radius=float(rng.uniform(0.5, 4)), sigma=float(rng.uniform(0.5, 2)),
                    neg_factor=float(rng.uniform(0.1, 0.9)))
    def make(geneo_num):
        gp, names = {}, []
        for kind, n in geneo_num.items():
            for i in range(n):
                gp[f"{kind}_{i}"] = rp(kind); names.append(f"{kind}_{i}")
        G = len(names)
        lam = {f"lambda_{n}": float(rng.uniform(-2 / G, 1 / G)) for n in names}
        return gp, lam
    gp, lam = make({"cy": 2, "cone": 1, "neg": 1})
    a, m = forward_case("c1_999", {"cy": 2, "cone": 1, "neg": 1}, (9, 9, 9), (20, 20, 20), 2, gp, lam,
                        "lambda_cone_0", seed=12, density=0.1)
    out.update(a); meta.update(m)
    # (3) full 16-kernel bank on a small grid
    gp, lam = make({"cy": 6, "cone": 5, "neg": 5})
    a, m = forward_case("g16_999", {"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9), (16, 16, 16), 1, gp, lam,
                        "lambda_cy_3", seed=13, density=0.15)
    out.update(a); meta.update(m)
    # (4) even kernel dims -> asymmetric 'same' padding; non-square footprint scramble
    gp, lam = make({"cy": 1, "cone": 2, "neg": 2})
    a, m = forward_case("even_656", {"cy": 1, "cone": 2, "neg": 2}, (6, 5, 6), (12, 14, 18), 2,
                        {k: ({**v, "apex": 2.0} if "apex" in v else v) for k, v in gp.items()}, lam,
                        "lambda_neg_1", seed=14, density=0.2)
    out.update(a); meta.update(m)
    np.savez_compressed(os.path.join(OUT, "geneo_forward.npz"), **out)
    import json
    with open(os.path.join(OUT, "geneo_forward_meta.json"), "w") as f:
        json.dump(meta, f, indent=0)
    print("geneo_forward.npz:", sorted({k.split('/')[0] for k in out}))


def dump_forward_v1():
    """SCENE_Net (v1 module, SCENE_Net.py:121-226) forward with explicit parameters."""
    from core.models.SCENE_Net import SCENE_Net
    torch.manual_seed(0)
    model = SCENE_Net({"cy": 2, "cone": 1, "neg": 1}, (9, 7, 7), device=torch.device("cpu"))
    gp = {"cy_0": dict(radius=2.5, sigma=1.8), "cy_1": dict(radius=1.0, sigma=0.7),
          "cone_0": dict(radius=2.0, sigma=1.4, apex=4.0, cone_radius=3.0, cone_inc=0.2),
          "neg_0": dict(radius=2.0, sigma=0.9, neg_factor=0.2)}
    lam = {"lambda_cy_0": 0.11, "lambda_cy_1": 0.27, "lambda_cone_0": 0.35, "lambda_neg_0": 0.05}
    set_model(model, gp, lam, "lambda_cy_1")
    rng = np.random.default_rng(31)
    x = (rng.random((2, 1, 14, 12, 20)) < 0.12).astype(np.float64)
    with torch.no_grad():
        kernels = torch.stack([model.geneos[g].compute_kernel() for g in model.geneos])
        conv = torch.nn.functional.conv3d(torch.from_numpy(x), kernels, padding="same")
        out = model(torch.from_numpy(x))
    names = list(model.geneos.keys())
    np.savez_compressed(os.path.join(OUT, "geneo_forward_v1.npz"), x=x.astype(np.uint8), bank=kernels.numpy(),
                        conv=conv.numpy(), out=out.numpy(), names=np.array(names),
                        lambdas=np.array([lam[f"lambda_{n}"] for n in names], dtype=np.float32),
                        last=np.array(names.index("cy_1")), kernel_size=np.array((9, 7, 7)))
    import json
    with open(os.path.join(OUT, "geneo_forward_v1_meta.json"), "w") as f:
        json.dump(dict(geneo_params=gp, names=names, state_dict_keys=list(model.state_dict().keys())), f, indent=0)
    print("geneo_forward_v1.npz")


def dump_module_contract():
    """State-dict keys / accessor names the drop-in must keep."""
    torch.manual_seed(3)
    model = SceneNet({"cy": 2, "cone": 1, "neg": 1}, (9, 9, 9))
    import json
    contract = dict(
        state_dict_keys=list(model.state_dict().keys()),
        named_parameters=[n for n, _ in model.named_parameters()],
        requires_grad={n: bool(p.requires_grad) for n, p in model.named_parameters()
                       if "lambda" not in n},
        geneo_param_names=list(model.get_geneo_params().keys()),
        in_dict_keys=list(model.get_model_parameters_in_dict().keys()),
        cvx_keys=list(model.get_cvx_coefficients().keys()),
        n_frozen_lambdas=sum(1 for n, p in model.named_parameters() if "lambda" in n and not p.requires_grad),
        # construction under torch.manual_seed(3): the drop-in makes the same RNG draws in the same order
        seed=3, seeded_values=model.get_model_parameters_in_dict(), seeded_last_lambda=model.last_lambda,
        num_total_params=model.get_num_total_params(),
    )
    with open(os.path.join(OUT, "module_contract.json"), "w") as f:
        json.dump(contract, f, indent=1)
    print("module_contract.json:", len(contract["state_dict_keys"]), "keys")


def dump_voxel_normalize():
    """normalize_xyz (pcd_processing.py:305-321) and ToFullDense (torch_transforms.py:17-40) run on
    integer count grids -- the two steps of the voxel path that do NOT depend on pyntcloud."""
    from utils import pcd_processing as eda
    from core.datasets.torch_transforms import ToFullDense
    rng = np.random.default_rng(21)
    out = {}
    cases = {
        "sparse": (rng.random((6, 5, 7)) < 0.2) * rng.integers(1, 40, (6, 5, 7)),
        "dense": rng.integers(0, 9, (4, 4, 4)),
        "fullcol": np.where(np.arange(5)[None, None, :] == 2, rng.integers(3, 9, (3, 4, 5)),
                            (rng.random((3, 4, 5)) < 0.5) * rng.integers(1, 5, (3, 4, 5))),  # column y=2 has min > 0
        "constcol": np.where(np.arange(5)[None, None, :] == 1, 4, rng.integers(0, 3, (3, 4, 5))),  # max == min > 0
        "empty": np.zeros((3, 3, 3), dtype=np.int64),
    }
    dens = ToFullDense(apply=(True, True))
    for k, c in cases.items():
        c = c.astype(np.float64)
        _, norm = eda.normalize_xyz(c.copy())
        out[f"{k}/counts"] = c
        out[f"{k}/norm"] = norm
        v, g = dens((torch.from_numpy(norm), torch.from_numpy(norm * 0.5)))
        out[f"{k}/dense"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "voxel_normalize.npz"), **out)
    print("voxel_normalize.npz:", list(cases))


def dump_vxg_to_xyz():
    """vxg_to_xyz (utils/voxelization.py:328-360): grid -> (V, 4) rows, default and explicit origin / voxel size."""
    from utils import voxelization as vox
    rng = np.random.default_rng(33)
    out = {}
    cases = {
        "default_f32": (torch.from_numpy(rng.random((3, 4, 5)).astype(np.float32)), None, None),
        "binary_f64": (torch.from_numpy((rng.random((4, 2, 6)) < 0.3).astype(np.float64)), None, None),
        "utm_f64": (torch.from_numpy(rng.random((5, 3, 2))), np.array([5.44e5 + 0.1, 4.634e6 + 0.7, 150.3]),
                    np.array([0.46875, 0.3, 0.937])),
        "int_origin": (torch.from_numpy(rng.random((2, 2, 2)).astype(np.float32)), np.array([3, -7, 11]),
                       np.array([2, 5, 1])),
    }
    for k, (g, o, vs) in cases.items():
        r = vox.vxg_to_xyz(g, origin=o, voxel_size=vs)
        out[f"{k}/grid"] = g.numpy()
        if o is not None:
            out[f"{k}/origin"] = o
            out[f"{k}/voxel_size"] = vs
        out[f"{k}/rows"] = np.asarray(r, dtype=np.float64)
    out["cases"] = np.array(list(cases))
    np.savez_compressed(os.path.join(OUT, "vxg_to_xyz.npz"), **out)
    print("vxg_to_xyz.npz:", list(cases))


def dump_real_tile_subset():
    """A ~6k-point subset of the reference's data-sample/sample_575.npy (TS40K tile, (N,4) f64 x,y,z,label at
    UTM scale): the six bbox-defining points, every tower (label 15) point and a random remainder."""
    a = np.load(os.path.join(REF, "data-sample", "sample_575.npy"))
    rng = np.random.default_rng(575)
    idx = set()
    for c in range(3):
        idx.add(int(a[:, c].argmin())); idx.add(int(a[:, c].argmax()))
    idx.update(np.where(a[:, 3] == 15)[0].tolist())
    idx.update(rng.choice(len(a), 6000 - len(idx), replace=False).tolist())
    sub = a[np.array(sorted(idx))[:6000]]
    np.save(os.path.join(OUT, "ts40k_sample575_subset.npy"), sub)
    print("ts40k_sample575_subset.npy:", sub.shape)


def dump_real_tile_full():
    """BASELINE C1's input verbatim: all 58 243 rows of data-sample/sample_575.npy ((N,4) f64 x,y,z,label, UTM scale),
    stored compressed.  A data file of the reference, not code."""
    a = np.load(os.path.join(REF, "data-sample", "sample_575.npy"))
    assert a.shape == (58243, 4) and a.dtype == np.float64
    np.savez_compressed(os.path.join(OUT, "ts40k_sample575_full.npz"), tile=a)
    print("ts40k_sample575_full.npz:", a.shape)


def dump_loss():
    """Losses and gradients of the reference's own criterion classes (core/criterions/geneo_loss.py) on small
    (pred, gt) pairs: GENEO_Loss, GENEO_Tversky_Loss, GENEO_Dice_Loss, plus WeightedMSE.get_weight_target and
    hist_frequency_estimation.  The classes write ./hist_estimation.pickle, so they are built in a scratch cwd."""
    import tempfile
    from core.criterions.geneo_loss import GENEO_Dice_Loss, GENEO_Loss, GENEO_Tversky_Loss
    from core.criterions.dice_loss import BinaryDiceLoss_BCE

    cases = {
        # name: (shape, dtype, gt kind, explicit freqs or None (estimate from gt), hyper-parameters)
        "ratio_f32_estimated": ((2, 1, 8, 10, 12), torch.float32, "ratio", None, dict()),
        "binary_f64_estimated": ((3, 1, 6, 8, 8), torch.float64, "binary", None,
                                 dict(weight_alpha=2.0, weight_epsilon=0.01, mse_weight=3.0, convex_weight=2.0)),
        "ratio_f64_large_freqs": ((2, 1, 8, 8, 8), torch.float64, "ratio",
                                  [5_000_000, 1200, 800, 700, 650, 400, 300, 310, 150, 9000],
                                  dict(tversky_alpha=0.3, tversky_beta=0.7, focal_gamma=2.0, tversky_smooth=0.5)),
        "binary_f32_chain_freqs": ((1, 1, 8, 8, 16), torch.float32, "binary", [40, 3, 5, 7, 9, 2, 4, 6, 8, 11],
                                   dict(weight_alpha=0.5, weight_epsilon=0.2, focal_gamma=1.5)),
    }
    out = {}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            for ci, (name, (shape, dt, kind, freqs, hp)) in enumerate(cases.items()):
                g = torch.Generator().manual_seed(100 + ci)
                u = torch.rand(shape, generator=g, dtype=torch.float64)
                if kind == "binary":
                    gt = (u < 0.08).to(dt)
                else:  # reg_on_voxel-like: mostly 0, some 1, some ratios in (0,1)
                    r = torch.rand(shape, generator=g, dtype=torch.float64)
                    gt = torch.where(u < 0.85, torch.zeros_like(r), torch.where(u < 0.92, torch.ones_like(r), r)).to(dt)
                pred0 = torch.rand(shape, generator=g, dtype=torch.float64).to(dt)
                names = ["lambda_cone_0", "lambda_cy_0", "lambda_cy_1", "lambda_neg_0"]
                lam = [0.41, -0.07, 0.9, -0.24]  # last (frozen) = 1 - sum(others) = -0.24; one negative trainable
                gpar = {"a_radius": 1.5, "b_sigma": -0.3, "c_neg_factor": 0.2, "d_cone_inc": -0.05}
                for cls_name, cls in [("geneo", GENEO_Loss), ("tversky", GENEO_Tversky_Loss), ("dice", GENEO_Dice_Loss)]:
                    crit = cls(targets=gt, weighting_scheme_path=None, **hp)
                    est_freqs, est_ranges = crit.freqs.clone(), crit.ranges.clone()
                    if freqs is not None:
                        crit.freqs = torch.tensor(freqs, dtype=torch.int64)
                    pred = pred0.clone().requires_grad_(True)
                    cvx = torch.nn.ParameterDict({n: torch.nn.Parameter(torch.tensor(v), requires_grad=(n != names[-1]))
                                                  for n, v in zip(names, lam)})
                    gp = torch.nn.ParameterDict({n: torch.nn.Parameter(torch.tensor(v)) for n, v in gpar.items()})
                    loss = cls.forward(crit, pred, gt, cvx, gp)
                    loss.backward()
                    out[f"{name}|{cls_name}|loss"] = loss.detach().numpy()
                    out[f"{name}|{cls_name}|grad_pred"] = pred.grad.numpy()
                    out[f"{name}|{cls_name}|grad_cvx"] = np.array(
                        [0.0 if cvx[n].grad is None else float(cvx[n].grad) for n in names])
                    out[f"{name}|{cls_name}|grad_params"] = np.array([float(gp[n].grad) for n in gpar])
                for red in ("mean", "sum"):   # the stand-alone weighted BCE + dice criterion ('dice_bce')
                    hp_w = {k: v for k, v in hp.items() if k in ("weight_alpha", "weight_epsilon", "mse_weight")}
                    crit = BinaryDiceLoss_BCE(targets=gt, weighting_scheme_path=None, reduction=red, **hp_w)
                    if freqs is not None:
                        crit.freqs = torch.tensor(freqs, dtype=torch.int64)
                    pred = pred0.clone().clamp(1e-4, 1 - 1e-4).requires_grad_(True)
                    loss = crit(pred, gt)
                    loss.backward()
                    out[f"{name}|dice_bce_{red}|loss"] = loss.detach().numpy()
                    out[f"{name}|dice_bce_{red}|grad_pred"] = pred.grad.numpy()
                out[f"{name}|pred"] = pred0.numpy()
                out[f"{name}|gt"] = gt.numpy()
                out[f"{name}|freqs"] = crit.freqs.numpy()
                out[f"{name}|ranges"] = crit.ranges.numpy()
                out[f"{name}|est_freqs"] = est_freqs.numpy()
                out[f"{name}|weights"] = crit.get_weight_target(gt).numpy()
                out[f"{name}|hp"] = np.array(repr(hp))
                out[f"{name}|cvx_names"] = np.array(names)
                out[f"{name}|cvx_values"] = np.array(lam, dtype=np.float32)
                out[f"{name}|param_names"] = np.array(list(gpar))
                out[f"{name}|param_values"] = np.array(list(gpar.values()), dtype=np.float32)
        finally:
            os.chdir(cwd)
    out["cases"] = np.array(list(cases))
    np.savez_compressed(os.path.join(OUT, "geneo_loss.npz"), **out)
    print("geneo_loss.npz:", list(cases))


def dump_loss_extra():
    """The criterion branches round 1 left out: BinaryDiceLoss with p != 2 and every reduction
    (core/criterions/dice_loss.py:33-51), BinaryDiceLoss_BCE with reduction='none' (:88-89; its `weights*bce + dice`
    only broadcasts for a batch of one).  Values and gradients w.r.t. the prediction, from the reference's classes."""
    import tempfile
    from core.criterions.dice_loss import BinaryDiceLoss, BinaryDiceLoss_BCE
    out = {}
    shapes = {"b3": (3, 1, 6, 8, 8), "b1": (1, 1, 8, 8, 16)}
    for si, (sname, shape) in enumerate(shapes.items()):
        g = torch.Generator().manual_seed(500 + si)
        for dt_name, dt in (("f32", torch.float32), ("f64", torch.float64)):
            gt = (torch.rand(shape, generator=g, dtype=torch.float64) < 0.1).to(dt)
            pred0 = torch.rand(shape, generator=g, dtype=torch.float64).clamp(1e-4, 1 - 1e-4).to(dt)
            out[f"{sname}_{dt_name}|pred"] = pred0.numpy()
            out[f"{sname}_{dt_name}|gt"] = gt.numpy()
            for p_ in (1, 2, 3):
                for red in ("mean", "sum", "none"):
                    for smooth in (1, 0.5):
                        crit = BinaryDiceLoss(smooth=smooth, p=p_, reduction=red)
                        pred = pred0.clone().requires_grad_(True)
                        loss = crit(pred, gt)
                        loss.sum().backward()
                        key = f"{sname}_{dt_name}|dice_p{p_}_{red}_s{smooth}"
                        out[key + "|loss"] = loss.detach().numpy()
                        out[key + "|grad_pred"] = pred.grad.numpy()
            if sname == "b1":
                cwd = os.getcwd()
                with tempfile.TemporaryDirectory() as tmp:
                    os.chdir(tmp)
                    try:
                        crit = BinaryDiceLoss_BCE(targets=gt, weighting_scheme_path=None, reduction="none",
                                                  weight_alpha=0.5, weight_epsilon=0.2)
                        crit.freqs = torch.tensor([40, 3, 5, 7, 9, 2, 4, 6, 8, 11], dtype=torch.int64)
                        pred = pred0.clone().requires_grad_(True)
                        loss = crit(pred, gt)
                        loss.sum().backward()
                        out[f"{sname}_{dt_name}|dice_bce_none|loss"] = loss.detach().numpy()
                        out[f"{sname}_{dt_name}|dice_bce_none|grad_pred"] = pred.grad.numpy()
                        out[f"{sname}_{dt_name}|dice_bce_none|freqs"] = crit.freqs.numpy()
                        out[f"{sname}_{dt_name}|dice_bce_none|ranges"] = crit.ranges.numpy()
                    finally:
                        os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "geneo_loss_extra.npz"), **out)
    print("geneo_loss_extra.npz:", len(out), "arrays")


if __name__ == "__main__":
    if sys.argv[1:] == ["loss_extra"]:
        dump_loss_extra()
        sys.exit(0)
    if sys.argv[1:] == ["loss"]:
        dump_loss()
        sys.exit(0)
    if sys.argv[1:] == ["tile"]:
        dump_real_tile_full()
        sys.exit(0)
    if sys.argv[1:] == ["vxg"]:
        dump_vxg_to_xyz()
        sys.exit(0)
    dump_loss()
    dump_loss_extra()
    dump_real_tile_subset()
    dump_real_tile_full()
    dump_vxg_to_xyz()
    dump_voxel_normalize()
    dump_kernels()
    dump_forward()
    dump_forward_v1()
    dump_module_contract()
