"""Host-side mirror of the reference's GENEO kernel classes (core/models/geneos/*.py).

Same class names, constructor arguments, static config helpers and error behaviour as
`cylinderv2` (cylinder.py:146-176), `arrow` (arrow.py:208-252) and `negSpherev2`
(neg_sphere.py:160-199); the kernel itself is built by the HIP bank builder
(sn_geneo_bank, csrc/bank.hip) -- there is no torch/CPU implementation here.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from . import _hip

KIND_OF_CLASS: Dict[str, int] = {}


def _as_f32_scalar(v, device) -> torch.Tensor:
    t = v if isinstance(v, torch.Tensor) else torch.tensor(v)
    return t.detach().to(device=device, dtype=torch.float32).reshape(())


def pack_params(kind: int, params: Dict[str, torch.Tensor], device) -> torch.Tensor:
    """One GENEO's scalars -> the [SN_NPARAM] fp32 slot vector of include/scenenet_hip.h."""
    zero = torch.zeros((), dtype=torch.float32, device=device)
    slots: List[torch.Tensor] = [zero] * _hip.SN_NPARAM
    slots[_hip.SN_P_RADIUS] = _as_f32_scalar(params["radius"], device)
    sigma = params.get("sigma")
    slots[_hip.SN_P_SIGMA] = _as_f32_scalar(1.0 if sigma is None else sigma, device)  # default sigma = 1
    if kind in (_hip.SN_GENEO_CONE, _hip.SN_GENEO_CONE_V1):
        slots[_hip.SN_P_APEX] = _as_f32_scalar(params["apex"], device)
        slots[_hip.SN_P_CONE_RADIUS] = _as_f32_scalar(params["cone_radius"], device)
        slots[_hip.SN_P_CONE_INC] = _as_f32_scalar(params["cone_inc"], device)
    elif kind in (_hip.SN_GENEO_NEG, _hip.SN_GENEO_NEG_V1):
        slots[_hip.SN_P_NEG_FACTOR] = _as_f32_scalar(params["neg_factor"], device)
    return torch.stack(slots)


def _kernel_device() -> torch.device:
    # GENEO_kernel_torch.py:30: kernels live on 'cuda' when available.  No CPU path here.
    if not torch.cuda.is_available():
        raise _hip.HipLibraryError("GENEO kernels are built by the HIP extension: no HIP device available")
    return torch.device("cuda", torch.cuda.current_device())


class GENEO_kernel_torch:
    """GENEO_kernel_torch.py:17-116.  Kernel shape is (z, x, y)."""

    KIND = -1

    def __init__(self, name, kernel_size, plot=False):
        self.name = name
        self.kernel_size = tuple(int(k) for k in kernel_size)
        self.plot = plot
        self.device = _kernel_device()
        self.volume = torch.prod(torch.tensor(self.kernel_size, device=self.device))
        self.kernel = self.compute_kernel()

    def _params(self) -> Dict[str, torch.Tensor]:
        raise NotImplementedError

    def compute_kernel(self) -> torch.Tensor:
        params = pack_params(self.KIND, self._params(), self.device).unsqueeze(0).contiguous()
        kinds = torch.tensor([self.KIND], dtype=torch.int32, device=self.device)
        status = torch.zeros(1, dtype=torch.int32, device=self.device)
        bank = _hip.geneo_bank(params, kinds, self.kernel_size, status)
        if self.KIND in (_hip.SN_GENEO_CONE, _hip.SN_GENEO_CONE_V1) and int(status.item()) != 0:
            # the reference fails in torch.tile / torch.stack with a wrong kernel depth (arrow.py:241-250)
            raise RuntimeError(f"arrow: int(apex) outside [0, {self.kernel_size[0]}]")
        return bank[0]

    def convolution(self, tensor: torch.Tensor, plot=False) -> torch.Tensor:
        """GENEO_kernel_torch.py:47-64: conv3d(tensor, kernel[None,None], padding='same'); tensor [B,1,Z,X,Y]."""
        act, _ = _hip.conv_bank(tensor.contiguous(), self.kernel.unsqueeze(0).contiguous(), None, want_act=True,
                                want_out=False)
        return act

    @staticmethod
    def mandatory_parameters():
        return []

    @staticmethod
    def geneo_parameters():
        return []

    @staticmethod
    def geneo_random_config(name="GENEO_rand"):
        # GENEO_kernel_torch.py:96-116 (no RNG draw: geneo_parameters() is empty at this level)
        return {"name": name, "kernel_size": (9, 9, 9), "plot": False, "geneo_params": {}, "non_trainable": []}


class cylinderv2(GENEO_kernel_torch):
    """cylinder.py:30-70 (ctor of cylinder_kernel) + :146-176."""

    KIND = _hip.SN_GENEO_CY

    def __init__(self, name, kernel_size, plot=False, **kwargs):
        if kwargs.get("radius") is None:
            raise KeyError("Provide a radius for the cylinder in the kernel.")
        self.radius = kwargs["radius"]
        self.sigma = kwargs["sigma"] if kwargs.get("sigma") is not None else 1
        super().__init__(name, kernel_size, plot)

    def _params(self):
        return {"radius": self.radius, "sigma": self.sigma}

    @staticmethod
    def mandatory_parameters():
        return ["radius"]

    @staticmethod
    def geneo_parameters():
        return cylinderv2.mandatory_parameters() + ["sigma"]

    @staticmethod
    def geneo_random_config(name="GENEO_rand"):
        cfg = GENEO_kernel_torch.geneo_random_config()
        cfg["geneo_params"] = {  # cylinder.py:115-118 (same RNG draw order)
            "radius": torch.randint(1, cfg["kernel_size"][1], (1,))[0] / 2,
            "sigma": torch.randint(5, 10, (1,))[0] / 5,
        }
        cfg["name"] = "cylinder"
        return cfg

    @staticmethod
    def geneo_smart_config(name="Smart_Cylinder"):
        return {"name": name, "kernel_size": (9, 6, 6), "plot": False, "non_trainable": [],
                "geneo_params": {"radius": torch.tensor(1.0), "sigma": torch.tensor(2.0)}}


class arrow(GENEO_kernel_torch):
    """arrow.py:30-113 (ctor of cone_kernel) + :208-252."""

    KIND = _hip.SN_GENEO_CONE

    def __init__(self, name, kernel_size, plot=False, **kwargs):
        if kwargs.get("radius") is None:
            raise KeyError("Provide a radius for the cylinder in the kernel.")
        if kwargs.get("apex") is None:
            raise KeyError("Provide a height for the cone.")
        if kwargs.get("cone_inc") is None:
            raise KeyError("Provide an inclination for the cone.")
        self.radius = kwargs["radius"]
        self.apex = kwargs["apex"]
        self.cone_inc = kwargs["cone_inc"]
        # arrow.py:84-87: default cone_radius = kernel_size[1] - 1
        self.cone_radius = kwargs["cone_radius"] if kwargs.get("cone_radius") is not None else float(kernel_size[1] - 1)
        self.sigma = kwargs["sigma"] if kwargs.get("sigma") is not None else 1
        super().__init__(name, kernel_size, plot)

    def _params(self):
        return {"radius": self.radius, "sigma": self.sigma, "apex": self.apex, "cone_radius": self.cone_radius,
                "cone_inc": self.cone_inc}

    @staticmethod
    def mandatory_parameters():
        return ["radius", "apex", "cone_radius", "cone_inc"]

    @staticmethod
    def geneo_parameters():
        return arrow.mandatory_parameters() + ["sigma"]

    @staticmethod
    def geneo_random_config(name="GENEO_rand"):
        cfg = GENEO_kernel_torch.geneo_random_config()
        k = cfg["kernel_size"]
        cfg["geneo_params"] = {  # arrow.py:123-129 (same RNG draw order)
            "radius": torch.randint(1, k[1], (1,))[0] / 2,
            "apex": torch.randint(int(k[0] / 2), k[0] - 1, (1,))[0],
            "cone_radius": torch.randint(1, k[1], (1,))[0] / 2,
            "cone_inc": torch.rand(1, )[0],
            "sigma": torch.randint(5, 10, (1,))[0] / 5,
        }
        cfg["name"] = "cone"
        cfg["non_trainable"] = ["apex"]
        return cfg

    @staticmethod
    def geneo_smart_config(name="Smart_Cylinder"):
        return {"name": name, "kernel_size": (9, 6, 6), "plot": False, "non_trainable": [],
                "geneo_params": {"radius": torch.tensor(1.0), "apex": torch.tensor(3.0),
                                 "cone_radius": torch.tensor(2.0), "cone_inc": torch.tensor(0.1),
                                 "sigma": torch.tensor(2.0)}}


class negSpherev2(GENEO_kernel_torch):
    """neg_sphere.py:29-78 (ctor of neg_sphere_kernel) + :160-199."""

    KIND = _hip.SN_GENEO_NEG

    def __init__(self, name, kernel_size, plot=False, **kwargs):
        if kwargs.get("radius") is None:
            raise KeyError("Provide a radius for the sphere.")
        if kwargs.get("neg_factor") is None:
            raise KeyError("Provide a negative factor for each sphere weight.")
        self.radius = kwargs["radius"]
        self.neg_factor = kwargs["neg_factor"]
        self.sigma = kwargs["sigma"] if kwargs.get("sigma") is not None else 1
        super().__init__(name, kernel_size, plot)

    def _params(self):
        return {"radius": self.radius, "sigma": self.sigma, "neg_factor": self.neg_factor}

    @staticmethod
    def mandatory_parameters():
        return ["radius", "neg_factor"]

    @staticmethod
    def geneo_parameters():
        return negSpherev2.mandatory_parameters() + ["sigma"]

    @staticmethod
    def geneo_random_config(name="GENEO_rand"):
        cfg = GENEO_kernel_torch.geneo_random_config()
        cfg["geneo_params"] = {  # neg_sphere.py:92-96 (same RNG draw order)
            "radius": torch.randint(1, cfg["kernel_size"][1], (1,))[0],
            "neg_factor": torch.randint(1, 10, (1,))[0] / 10,
            "sigma": torch.randint(5, 10, (1,))[0] / 10,
        }
        cfg["non_trainable"] = []
        cfg["name"] = "neg"
        return cfg

    @staticmethod
    def geneo_smart_config(name="Smart_Neg_Sphere"):
        return {"name": name, "kernel_size": (9, 6, 6), "plot": False, "non_trainable": [],
                "geneo_params": {"radius": torch.tensor(3.0), "sigma": torch.tensor(2.0),
                                 "neg_factor": torch.tensor(0.5)}}


KIND_OF_CLASS.update({"cy": cylinderv2.KIND, "cone": arrow.KIND, "neg": negSpherev2.KIND})
CLASS_OF_KEY = {"cy": cylinderv2, "cone": arrow, "neg": negSpherev2}  # SCENE_Net.py:259-272


# --------------------------------------------------------------------------- #
# v1 generators used by the v1 module `SCENE_Net` (core/models/SCENE_Net.py:158-170)
# --------------------------------------------------------------------------- #
class cylinder_kernel(cylinderv2):
    """cylinder.py:30-140: ring gaussian exp((|p-c|^2 - radius^2)^2 / (-2 sigma^2)), no sigma prefactor."""

    KIND = _hip.SN_GENEO_CY_V1


class cone_kernel(arrow):
    """arrow.py:30-205: cylinder below, ring gaussians of shrinking sigma_h = cone_radius sin(cone_inc pi / (2+h)) above;
    cone_inc is not clamped."""

    KIND = _hip.SN_GENEO_CONE_V1


class neg_sphere_kernel(negSpherev2):
    """neg_sphere.py:29-158: 3-D ring gaussian, sum_zero, minus neg_factor."""

    KIND = _hip.SN_GENEO_NEG_V1


CLASS_OF_KEY_V1 = {"cy": cylinder_kernel, "cone": cone_kernel, "neg": neg_sphere_kernel}  # SCENE_Net.py:158-170
KIND_OF_CLASS.update({"cy_v1": cylinder_kernel.KIND, "cone_v1": cone_kernel.KIND, "neg_v1": neg_sphere_kernel.KIND})
