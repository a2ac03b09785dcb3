"""Host-side mirror of core/models/SCENE_Net.py: `GENEO_Layer` (:56-113) and `SceneNet` (:229-339).

Same constructor signatures, attribute names, state-dict keys, accessors, RNG draw order at
construction and forward side effects as the reference, so it drops into
core/lit_modules/lit_model_wrappers.py:155 unchanged.  The forward itself is two HIP calls:
sn_geneo_bank (K2) and sn_conv_bank (K3, conv + convex head fused).  No CPU path.
"""
from __future__ import annotations

import weakref
from typing import Mapping, Optional, Tuple

import torch
import torch.nn as nn

from . import _hip
from .geneos import CLASS_OF_KEY, CLASS_OF_KEY_V1, GENEO_kernel_torch, pack_params


_SLOT_OF = {"radius": _hip.SN_P_RADIUS, "sigma": _hip.SN_P_SIGMA, "apex": _hip.SN_P_APEX,
            "cone_radius": _hip.SN_P_CONE_RADIUS, "cone_inc": _hip.SN_P_CONE_INC, "neg_factor": _hip.SN_P_NEG_FACTOR}


# ---- writes that no `_version` sees
# Everything this module keeps between calls (packed parameters, effective coefficients, K3L's tables, learnt guard verdicts)
# is keyed on each parameter's identity and `_version`.  Every in-place op and every re-assignment moves those -- a write
# through `.data` does not (`p.data` is a fresh tensor with its own version counter).  So the model's parameters are
# instances of a Parameter subclass whose `.data` property counts its uses: `p.data.fill_(..)`, `p.data = t`, an old-style
# `p.data.add_(-lr * g)` all pass through the getter or the setter, the counter is part of every cache key, and the next
# forward rebuilds what it kept.  (A READ through `.data` -- the reference's own `val.data.item()` in its logging -- also
# counts: a spurious rebuild costs microseconds.)  Not caught: a tensor obtained from `.data` EARLIER and written to later;
# a write through a raw pointer (CapturedTrainingStep.replay() calls invalidate_caches() itself).  `invalidate_caches()` is
# the explicit form.  The counter is global on purpose: nothing has to be attached to a parameter, so deepcopy / pickle /
# load_state_dict keep working, and a Parameter that lost the subclass on the way (pickle restores a plain nn.Parameter) is
# re-tagged the next time the keys are built.
_DATA_TOUCHES = [0]
_TENSOR_DATA = torch.Tensor.data


class _TrackedParameter(nn.Parameter):
    @property
    def data(self):
        _DATA_TOUCHES[0] += 1
        return _TENSOR_DATA.__get__(self)

    @data.setter
    def data(self, value):
        _DATA_TOUCHES[0] += 1
        _TENSOR_DATA.__set__(self, value)

    def __repr__(self):   # prints like the nn.Parameter it stands for
        t = torch.Tensor.detach(self).as_subclass(torch.Tensor)
        return "Parameter containing:\n" + repr(t.requires_grad_(self.requires_grad))


# torch's optimisers take their multi-tensor ("foreach" / fused) paths only for parameters whose exact type is on a list of
# two -- torch.Tensor and nn.Parameter; any other type drops them to one tiny kernel per parameter and operation
# ([measured] round 4: the captured training step 0.225 -> 0.309 ms, 14 extra elementwise launches per step, the moment
# the parameters carried the subclass).  The subclass adds a Python property and nothing a foreach kernel can see.
try:
    from torch.optim import optimizer as _torch_optimizer
    if _TrackedParameter not in _torch_optimizer._foreach_supported_types:
        _torch_optimizer._foreach_supported_types.append(_TrackedParameter)
except (ImportError, AttributeError):   # a torch without that list: the optimiser still works, one kernel per parameter
    pass


def _track(p):
    """tags a Parameter so that its `.data` accesses are counted (a class swap: same object, same storage)"""
    if type(p) is nn.Parameter:
        p.__class__ = _TrackedParameter
    return p


def _set_data_untracked(p, value) -> None:
    """p.data = value from inside this module (the flat buffer's aliasing): not a user write"""
    _TENSOR_DATA.__set__(p, value)


class _RiderOutputs(tuple):
    """(bank, lam) of SceneNet.train_rider() -- still a 2-tuple for callers that unpack it -- with the buffer set's identity
    and generation, so that the forward can tell its backward which buffers it read (see train_rider)."""

    def __new__(cls, pair, ring, k, gen):
        self = super().__new__(cls, pair)
        self.ring, self.k, self.gen = ring, k, gen
        return self


class _LivePack:
    """The packed-parameter tensor P of the latest differentiable forward.  The criteria (criterions.py) take their
    penalties from P, so every scalar receives ONE gradient from ONE autograd node instead of being stacked again and
    accumulated ~50 times.  Never copied or pickled with the module."""

    def __init__(self):
        self.packed = None        # weakref to P [G*SN_NPARAM + G] f32 (non-leaf): lives as long as that forward's graph
        self.versions = None      # parameter versions P was gathered at
        self.leaves = ()          # the nn.Parameters, in slot order
        self.mask_params = None   # device int8 [N]: 1 on slots that hold a GENEO parameter
        self.mask_cvx = None      # device int8 [N]: 2 on slots of the trainable convex coefficients
        self.mask_all = None      # both

    def current(self):
        """P if no parameter changed since it was gathered, else None."""
        P = self.packed() if self.packed is not None else None
        if P is None or self.versions != tuple(p._version for p in self.leaves):
            return None
        return P

    def __deepcopy__(self, memo):
        return _LivePack()

    def __reduce__(self):
        return (_LivePack, ())


class _GatherParamsFn(torch.autograd.Function):
    """leaves (0-dim nn.Parameters aliasing `flat`) -> P = copy of flat.  Backward hands every leaf its slot of dP
    as a view: no per-parameter kernels in either direction."""

    @staticmethod
    def forward(ctx, flat, slots, *leaves):
        ctx.slots = slots
        return flat.clone()

    @staticmethod
    def backward(ctx, gP):
        cols = gP.unbind(0)  # one call, N zero-dim views
        grads = tuple(cols[i] if need else None for i, need in zip(ctx.slots, ctx.needs_input_grad[2:]))
        return (None, None) + grads


class GENEO_Layer(nn.Module):
    """SCENE_Net.py:56-113."""

    def __init__(self, geneo_class: GENEO_kernel_torch, kernel_size: tuple = None, smart=False):
        super().__init__()
        self.geneo_class = geneo_class
        self.init_from_config(smart)
        if kernel_size is not None:
            self.kernel_size = kernel_size

    def init_from_config(self, smart=False):
        config = self.geneo_class.geneo_smart_config() if smart else self.geneo_class.geneo_random_config()
        self.name = config["name"]
        self.kernel_size = config["kernel_size"]
        self.plot = config["plot"]
        params = {}
        for param in config["geneo_params"]:
            t_param = torch.as_tensor(config["geneo_params"][param]).clone().detach().to(torch.float)
            params[param] = nn.Parameter(t_param, requires_grad=param not in config["non_trainable"])
        self.geneo_params = nn.ParameterDict(params)  # plain dict -> keys sorted, like the reference

    def init_from_kwargs(self, kernel_size, kwargs):
        self.kernel_size = kernel_size
        self.name = "GENEO"
        self.plot = False
        params = {}
        for param in self.geneo_class.mandatory_parameters():
            params[param] = nn.Parameter(torch.tensor(kwargs[param], dtype=torch.float))
        self.geneo_params = nn.ParameterDict(params)

    def compute_kernel(self) -> torch.Tensor:
        """[1, kz, kx, ky] float64 (SCENE_Net.py:103-106), built on the HIP device."""
        geneo = self.geneo_class(self.name, self.kernel_size, plot=self.plot, **self.geneo_params)
        kernel = geneo.kernel.to(dtype=torch.double)
        return kernel.view(1, *kernel.shape)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """conv3d(x, kernel, padding='same') -> [B,1,Z,X,Y] (SCENE_Net.py:108-113; the reference's own
        version reads an undefined self.device)."""
        geneo = self.geneo_class(self.name, self.kernel_size, plot=self.plot, **self.geneo_params)
        return geneo.convolution(x)


class SceneNet(nn.Module):
    """SCENE_Net.py:229-339."""

    GENEO_CLASSES = CLASS_OF_KEY
    # Binary-occupancy inputs whose bank activations are not asked for go through linearity (sn_conv_fused: one
    # combined kernel sum_i lambda_i K_i, ~4x faster, same output to ~5e-6); set False to force the 16-kernel
    # contraction (sn_conv_bank) everywhere.
    fused_forward = True
    # Storage dtype of the forward output where the linear forward (sn_conv_fused) serves the shape: None = the input's
    # float dtype (f32 for byte grids); torch.bfloat16 = bf16 activations for training (the reference's `precision:
    # 16`, defaults_config.yml:83-84): the prediction, its gradient from the criterion and what the backward
    # correlation reads are then 2 bytes per voxel; every sum stays fp32 / fp64.
    activation_dtype = None
    # lambda init range as a function of the number of GENEOs (SCENE_Net.py:276-277)
    LAMBDA_RANGE = staticmethod(lambda n: (-2 / n, 1 / n))

    def __init__(self, geneo_num=None, kernel_size=None, plot=False):
        super().__init__()
        self.sizes = {"cy": 1, "cone": 1, "neg": 1} if geneo_num is None else geneo_num
        if kernel_size is not None:
            self.kernel_size = kernel_size
        self.geneos: Mapping[str, GENEO_Layer] = nn.ModuleDict()
        for key in self.sizes:
            if key in self.GENEO_CLASSES:
                for i in range(self.sizes[key]):
                    self.geneos[f"{key}_{i}"] = GENEO_Layer(self.GENEO_CLASSES[key], kernel_size=kernel_size)

        # --- convex coefficients (SCENE_Net.py:274-293), same RNG draws
        num_lambdas = sum(self.sizes.values())
        lambda_init_min, lambda_init_max = self.LAMBDA_RANGE(num_lambdas)
        lambdas = (lambda_init_max - lambda_init_min) * torch.rand(num_lambdas, dtype=torch.float) + lambda_init_min
        lambdas = [nn.Parameter(lamb) for lamb in lambdas]
        self.lambda_names = [f"lambda_{key}_{i}" for key, val in self.sizes.items() for i in range(val)]
        self.last_lambda = self.lambda_names[torch.randint(0, num_lambdas, (1,))[0]]
        if plot:
            print(f"last cvx_coeff: {self.last_lambda}")
        d = dict(zip(self.lambda_names, lambdas))  # last cvx_coeff is 1 - sum(others)
        d[self.last_lambda] = nn.Parameter(1 - sum(d.values()) + d[self.last_lambda], requires_grad=False)
        self.lambdas_dict = nn.ParameterDict(d)
        self._pack_cache = None
        self._lambda_cache = None
        self._flat = None          # training path: one fp32 buffer every nn.Parameter aliases (see _flat_sync)
        self._flat_meta = None
        self._live = _LivePack()
        if plot:
            print(f"Total Number of train params = {self.get_num_total_params()}")

    # ------------------------------------------------------------------ accessors (SCENE_Net.py:299-319)
    def get_cvx_coefficients(self):
        object.__setattr__(self.lambdas_dict, "_sn_live", self._live)
        return self.lambdas_dict

    def invalidate_caches(self) -> None:
        """Drops everything derived from the parameters that is kept between calls (packed parameters, effective
        coefficients, the fused forward's tables, the learnt guard verdicts).  The caches key on each parameter's identity
        and `_version` (every in-place op, every re-assignment) and on the count of `.data` accesses (_TrackedParameter: a
        write through `.data` is seen too).  What is left for this call: a write through a raw pointer -- a replayed
        optimiser graph (CapturedTrainingStep.replay() calls it) -- or through a `.data` tensor taken earlier."""
        self._pack_cache = None
        self._lambda_cache = None
        for k in ("_fused_state", "_prepared_verdict", "_geneo_params_cache"):
            self.__dict__.pop(k, None)

    def get_num_total_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def get_model_parameters(self, detach=False):
        if detach:
            return {name: param.detach().clone() for name, param in self.named_parameters()}
        return {name: param for name, param in self.named_parameters()}

    def _geneo_leaves(self):
        """The GENEO nn.Parameters in ModuleDict order (identity tuple: cheap to compare between calls)."""
        return tuple(p for layer in self.geneos.values() for p in layer.geneo_params._parameters.values())

    def get_geneo_params(self):
        leaves = self._geneo_leaves()
        cached = self.__dict__.get("_geneo_params_cache")
        if cached is not None and len(cached[0]) == len(leaves) and all(a is b for a, b in zip(cached[0], leaves)):
            return cached[1]   # same Parameter objects as last time: same dict (the reference rebuilds it per call)
        d = nn.ParameterDict(dict([(name.replace(".", "_"), p) for name, p in self.named_parameters()
                                   if "lambda" not in name]))
        object.__setattr__(d, "_sn_live", self._live)
        self.__dict__["_geneo_params_cache"] = (leaves, d)
        return d

    def get_model_parameters_in_dict(self):
        ddd = {}
        for key, val in self.named_parameters():
            key_split = key.split(".")
            parameter_name = f"{key_split[-3]}.{key_split[-1]}" if "geneo" in key else key_split[-1]
            ddd[parameter_name] = val.data.item()
        return ddd

    # ------------------------------------------------------------------ host logic
    def kernel_size_of_bank(self) -> Tuple[int, int, int]:
        layers = tuple(self.geneos._modules.values())
        cached = self.__dict__.get("_ks_cache")
        if cached is not None and len(cached[0]) == len(layers) and all(a is b for a, b in zip(cached[0], layers)) \
                and all(l.kernel_size is k for l, k in zip(layers, cached[1])):
            return cached[2]
        sizes = {tuple(int(k) for k in layer.kernel_size) for layer in layers}
        if len(sizes) != 1:
            raise RuntimeError(f"GENEO kernels of different sizes cannot be stacked: {sizes}")  # torch.stack fails
        ks = next(iter(sizes))
        self.__dict__["_ks_cache"] = (layers, tuple(l.kernel_size for l in layers), ks)
        return ks

    def packed_params(self, device) -> Tuple[torch.Tensor, torch.Tensor]:
        """([G, SN_NPARAM] f32, [G] i32) on `device`; re-packed only when a parameter changed."""
        # (the key walks the plain dicts behind ModuleDict / ParameterDict: through their public iterators this one line
        # cost ~30 us of the eager step's ~150 us of host time)
        key = (device, _DATA_TOUCHES[0]) + tuple((id(_track(p)), p._version) for p in self._geneo_leaves())
        if self._pack_cache is not None and self._pack_cache[0] == key:
            return self._pack_cache[1], self._pack_cache[2]
        rows, kinds = [], []
        for layer in self.geneos.values():
            kind = layer.geneo_class.KIND
            for m in layer.geneo_class.mandatory_parameters():
                if m not in layer.geneo_params:
                    raise KeyError(f"GENEO {layer.name}: missing mandatory parameter {m}")
            rows.append(pack_params(kind, layer.geneo_params, device))
            kinds.append(kind)
            if kind in (_hip.SN_GENEO_CONE, _hip.SN_GENEO_CONE_V1):
                hc = int(layer.geneo_params["apex"].detach().to(torch.int).item())
                if hc < 0 or hc > int(layer.kernel_size[0]):
                    raise RuntimeError(f"arrow: int(apex)={hc} outside [0, {layer.kernel_size[0]}]")
        params = torch.stack(rows).contiguous()
        kinds_t = torch.tensor(kinds, dtype=torch.int32, device=device)
        self._pack_cache = (key, params, kinds_t)
        return params, kinds_t

    def effective_lambdas(self, device) -> torch.Tensor:
        """[G] f32 in GENEO order; the `last_lambda` entry is 1 - sum(lambdas_dict.values()) + last
        (SCENE_Net.py:331), summed in ParameterDict order like the reference.  Also performs the
        reference's side effect of re-creating lambdas_dict[last_lambda] (SCENE_Net.py:333)."""
        # ~17 tiny dependent device ops: redone only when a coefficient (or last_lambda) changed since the last
        # call -- 1 - sum(others) is then already what lambdas_dict[last_lambda] holds.
        key = (device, self.last_lambda, _DATA_TOUCHES[0]) + tuple((id(_track(p)), p._version)
                                                                  for p in self.lambdas_dict._parameters.values())
        if self._lambda_cache is not None and self._lambda_cache[0] == key:
            return self._lambda_cache[1]
        last = 1 - sum(self.lambdas_dict.values()) + self.lambdas_dict[self.last_lambda]
        self.lambdas_dict[self.last_lambda] = nn.Parameter(last.detach(), requires_grad=False)
        vals = [self.lambdas_dict[f"lambda_{g}"].detach() for g in self.geneos]
        lam = torch.stack(vals).to(device=device, dtype=torch.float32).contiguous()
        key = (device, self.last_lambda, _DATA_TOUCHES[0]) + tuple((id(_track(p)), p._version)
                                                                  for p in self.lambdas_dict._parameters.values())
        self._lambda_cache = (key, lam)
        return lam

    def compute_bank(self, device=None) -> torch.Tensor:
        """[G, kz, kx, ky] f32 on the HIP device (the stack at SCENE_Net.py:324, before the fp64 cast)."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        params, kinds = self.packed_params(device)
        return _hip.geneo_bank(params, kinds, self.kernel_size_of_bank())

    def compute_bank_prepared(self, device=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """(bank [G,9,9,9] f32, prep uint8): compute_bank and the int8 contraction's per-bank preparation in ONE launch
        (sn_geneo_bank_prep), written into two buffers the model keeps per device -- nothing is allocated per call, so
        the launch can sit on a side stream next to the voxelisation (ScenePipeline.bank_beside; bank_rider() is the form
        without any launch of its own).  9 x 9 x 9 kernels only.  The
        reference rebuilds its kernels at every forward (SCENE_Net.py:322-327); so does this."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.kernel_size_of_bank() != (9, 9, 9):
            raise _hip.HipLibraryError("compute_bank_prepared serves 9 x 9 x 9 kernels")
        params, kinds = self.packed_params(device)
        G = params.shape[0]
        bufs = self.__dict__.get("_prepared_bufs")
        if bufs is None or bufs[0].device != device or bufs[0].shape[0] != G:
            bufs = (torch.empty((G, 9, 9, 9), dtype=torch.float32, device=device),
                    torch.zeros(_hip.SN_CONV_PREP_BYTES * ((G + 15) // 16), dtype=torch.uint8, device=device))
            self.__dict__["_prepared_bufs"] = bufs
        bank, _, prep = _hip.geneo_bank_prep(params, kinds, bank=bufs[0], prep=bufs[1])
        return bank, prep

    def bank_rider(self, device=None):
        """(params, kinds, bank, prep): what compute_bank_prepared would launch, as arguments for K1's first launch
        (voxelize_batch(bank_rider=...) -> sn_voxel_occupancy_fused_bank): the bank and the preparation blob are then
        written by extra workgroups of the bounding-box kernel -- no launch, no stream fork, no event for K2.  The two
        buffers are the model's persistent ones (compute_bank_prepared's)."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.kernel_size_of_bank() != (9, 9, 9):
            raise _hip.HipLibraryError("bank_rider serves 9 x 9 x 9 kernels")
        params, kinds = self.packed_params(device)
        G = params.shape[0]
        bufs = self.__dict__.get("_prepared_bufs")
        if bufs is None or bufs[0].device != device or bufs[0].shape[0] != G:
            bufs = (torch.empty((G, 9, 9, 9), dtype=torch.float32, device=device),
                    torch.zeros(_hip.SN_CONV_PREP_BYTES * ((G + 15) // 16), dtype=torch.uint8, device=device))
            self.__dict__["_prepared_bufs"] = bufs
        return params, kinds, bufs[0], bufs[1]

    def train_rider(self, device=None):
        """The opener of a TRAINING forward -- bank + effective coefficients, sn_geneo_bank_lambdas -- as riders of the
        voxelisation's first launch: returns (rider, (bank, lam)); hand `rider` to voxelize_batch(bank_rider=...) and
        `(bank, lam)` to forward(x, bank_lam=...) of the same step.  Reads the flat parameter buffer the nn.Parameters
        alias (as the autograd function would) and refreshes the frozen coefficient in place.  9 x 9 x 9 banks."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.kernel_size_of_bank() != (9, 9, 9):
            raise _hip.HipLibraryError("train_rider serves 9 x 9 x 9 kernels")
        flat, meta, _ = self._flat_sync(device)
        G = meta["G"]
        n = G * _hip.SN_NPARAM
        # The rider writes (bank, lam) through raw pointers into buffers the model keeps -- no allocation per step -- and the
        # forward saves them for its backward.  A LATER rider launch would rewrite them in place with no version bump, so
        # autograd's saved-tensor check could not see it.  Two buffer sets, used alternately, each with a
        # generation number: the backward of a forward whose set has been handed out again since raises instead of reading
        # another step's bank.  (One forward may overlap -- a validation pass with grad enabled, retain_graph; a third
        # before the first one's backward is refused loudly.)  No copy, no launch.
        ring = self.__dict__.get("_train_rider_bufs")
        if ring is None or ring["sets"][0][0].device != device or ring["sets"][0][0].shape[0] != G:
            ring = {"sets": [(torch.empty((G, 9, 9, 9), dtype=torch.float32, device=device),
                              torch.zeros(_hip.SN_CONV_PREP_BYTES * ((G + 15) // 16), dtype=torch.uint8, device=device),
                              torch.empty(G, dtype=torch.float32, device=device)) for _ in range(2)],
                    "gen": [0, 0], "next": 0}
            self.__dict__["_train_rider_bufs"] = ring
        k = ring["next"]
        ring["next"] = 1 - k
        ring["gen"][k] += 1
        bank, prep, lam = ring["sets"][k]
        rider = (flat[:n].view(G, _hip.SN_NPARAM), meta["kinds"], bank, prep, flat[n:], meta["order"], meta["last"], lam)
        return rider, _RiderOutputs((bank, lam), ring, k, ring["gen"][k])

    def contract_prepared(self, x: torch.Tensor, bank: torch.Tensor, lam: torch.Tensor, prep: torch.Tensor,
                          want_act: bool = False, out_dtype: Optional[torch.dtype] = None):
        """sn_conv_bank_prepared on (bank, lam, prep) of THIS model's current parameters -> (act | None, out).  The walk's
        verdict (served / fp32 form / unfolded body) depends on the weights, the coefficients, the tolerance and the
        outputs asked for: once it has been read back as "served" for the current parameter versions -- asynchronously,
        no synchronisation (_hip.PreparedVerdict) -- the empty fallback launch behind the walk is left out, until a
        parameter changes."""
        key = (x.device, bool(want_act), _hip.get_option("conv_i8_tolerance_ppb"),
               self._pack_cache[0] if self._pack_cache is not None else None,
               self._lambda_cache[0] if self._lambda_cache is not None else None, prep.data_ptr())
        verdict = self.__dict__.setdefault("_prepared_verdict", _hip.PreparedVerdict())
        served = verdict.served(key)
        res = _hip.conv_bank(x, bank, lam, want_act=want_act, want_out=True, out_dtype=out_dtype, prep=prep,
                             assume_served=served)
        if not served:
            verdict.note(prep, key)
        return res

    def fused_served(self, x: torch.Tensor, bank: torch.Tensor, lam: torch.Tensor, out_dtype: torch.dtype):
        """sn_conv_fused on (bank, lam) of THIS model's current parameters, with what depends on them alone taken out of
        the call: the kernel's tables (K*, its fixed point, the digit tables: ~8 us that every workgroup otherwise builds
        for itself) are prepared into a blob once per parameter version (sn_conv_fused_prep -- like effective_lambdas'
        cache; not inside a graph capture, which outlives the versions), and the guard's verdict is learnt as
        in contract_prepared (asynchronously, keyed on the parameter versions and the tolerance): once it has read
        "served" the gated fp32 launches behind the combined kernel are left out."""
        ks = tuple(int(k) for k in bank.shape[1:])
        key = (x.device, _hip.get_option("conv_i8_tolerance_ppb"),
               self._pack_cache[0] if self._pack_cache is not None else None,
               self._lambda_cache[0] if self._lambda_cache is not None else None)
        state = self.__dict__.get("_fused_state")
        if state is None or state["blob"].device != x.device or state["ks"] != ks:
            state = {"verdict": _hip.PreparedVerdict(), "ks": ks, "key": None,
                     "blob": torch.empty(_hip.conv_fused_prep_bytes(ks), dtype=torch.uint8, device=x.device)}
            self.__dict__["_fused_state"] = state
        if torch.cuda.is_current_stream_capturing():
            # a captured graph outlives these parameter versions: the kernel builds its tables itself, the fallback stays
            return _hip.conv_fused(x, bank, lam, out_dtype=out_dtype)
        if state["key"] != key:
            _hip.conv_fused_prep(bank, lam, state["blob"])
            state["key"] = key
        served = state["verdict"].served(key)
        out = _hip.conv_fused(x, bank, lam, out_dtype=out_dtype, prep=state["blob"], assume_served=served)
        if not served and out_dtype != torch.bfloat16:   # (bf16 output runs unguarded: nothing to learn)
            state["verdict"].note_words(_hip.conv_fused_prep_verdict(state["blob"], ks), key)
        return out

    def serves_prepared(self, x: torch.Tensor) -> bool:
        """binary occupancy and a 9 x 9 x 9 bank: the z-walk kernel behind sn_conv_bank_prepared"""
        return x.dtype == torch.bool and x.is_cuda and self.kernel_size_of_bank() == (9, 9, 9)

    # ------------------------------------------------------------------ differentiable host logic (training)
    def _leaf_slots(self):
        """[(nn.Parameter, slot)]: GENEO g's parameters at g*SN_NPARAM + slot, coefficient g at G*SN_NPARAM + g."""
        G = len(self.geneos)
        leaves = []
        for g, layer in enumerate(self.geneos.values()):
            for name, p in layer.geneo_params.items():
                if name not in _SLOT_OF:
                    raise KeyError(f"GENEO {layer.name}: unknown parameter {name}")
                leaves.append((p, g * _hip.SN_NPARAM + _SLOT_OF[name]))
        for g, n in enumerate(self.geneos):
            leaves.append((self.lambdas_dict[f"lambda_{n}"], G * _hip.SN_NPARAM + g))
        return leaves

    def _flat_sync(self, device):
        """One fp32 device buffer [G*SN_NPARAM + G] that every nn.Parameter of the model aliases (p.data is a 0-dim
        view of its slot), so packing for the kernels costs nothing and an optimiser step updates the buffer in
        place.  Re-established (one stack + one scatter) whenever something replaced a parameter's storage
        (.to(), a fresh last-lambda Parameter from the inference path, ...)."""
        flat, meta = self._flat, self._flat_meta
        if flat is not None and flat.device == device and meta["last_lambda"] == self.last_lambda:
            # same Parameter objects (identity walk over the dicts), each still a view of its slot
            current = self._geneo_leaves() + tuple(self.lambdas_dict._parameters[n] for n in meta["lambda_names"])
            leaves, base = meta["leaves"], flat.data_ptr()
            if len(current) == len(leaves) and all(
                    a is p and p.data_ptr() == base + 4 * i for a, (p, i) in zip(current, leaves)):
                return flat, meta, leaves
        leaves = self._leaf_slots()
        self.packed_params(device)  # validation: mandatory parameters, apex range (cached on parameter versions)
        G = len(self.geneos)
        vals = torch.stack([p.detach().to(device=device, dtype=torch.float32) for p, _ in leaves])
        flat = torch.zeros(G * _hip.SN_NPARAM + G, dtype=torch.float32, device=device)
        flat[torch.arange(G, device=device) * _hip.SN_NPARAM + _hip.SN_P_SIGMA] = 1.0  # default sigma = 1
        slots = torch.tensor([i for _, i in leaves], dtype=torch.int64, device=device)
        flat[slots] = vals
        for p, i in leaves:
            _set_data_untracked(p, flat[i])
        names = list(self.geneos)
        order = sorted(range(G), key=lambda i: f"lambda_{names[i]}")  # nn.ParameterDict order (sorted names)
        last = names.index(self.last_lambda.replace("lambda_", "", 1))
        n_geneo = len(leaves) - G
        self._flat = flat
        self._flat_meta = {
            "last_lambda": self.last_lambda, "last": last, "G": G, "leaves": leaves,
            "lambda_names": tuple(f"lambda_{n}" for n in names),
            "order": torch.tensor(order, dtype=torch.int32, device=device),
            "kinds": torch.tensor([l.geneo_class.KIND for l in self.geneos.values()], dtype=torch.int32, device=device),
            "slots": tuple(i for _, i in leaves),
        }
        mp = torch.zeros(flat.numel(), dtype=torch.int8)
        mp[[i for _, i in leaves[:n_geneo]]] = 1
        mc = torch.zeros(flat.numel(), dtype=torch.int8)
        mc[[G * _hip.SN_NPARAM + g for g in range(G) if g != last]] = 2
        self._flat_meta.update(mask_params=mp.to(device), mask_cvx=mc.to(device), mask_all=(mp + mc).to(device))
        return flat, self._flat_meta, leaves

    def forward(self, x: torch.Tensor, return_bank_activations: bool = False, bank_lam=None):
        """x [B,1,Z,X,Y] on a HIP device -> relu(tanh(sum_i lambda_i conv3d(x, K_i))) [B,1,Z,X,Y], same dtype
        (f32 for bool / u8 input).  With return_bank_activations=True also returns conv [B,G,Z,X,Y]
        (SCENE_Net.py:325; not differentiable).  Under autograd the output carries the graph to every trainable
        scalar: backward = sn_conv_corr + sn_geneo_bank_bwd.  bank_lam: (bank, lam) of train_rider() -- built from the
        current parameters by this step's voxelisation launch -- instead of the forward's own opener."""
        if not x.is_cuda:
            raise _hip.HipLibraryError("SceneNet.forward runs on the HIP device only (no CPU fallback): move x to cuda")
        ks = self.kernel_size_of_bank()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            flat, meta, leaves = self._flat_sync(x.device)
            params = [p for p, _ in leaves]
            P = _GatherParamsFn.apply(flat, meta["slots"], *params)
            live = self._live
            live.packed, live.leaves = weakref.ref(P), tuple(params)
            live.mask_params, live.mask_cvx, live.mask_all = meta["mask_params"], meta["mask_cvx"], meta["mask_all"]
            out, act = _GeneoForwardFn.apply(x.contiguous(), P, flat, meta, ks, return_bank_activations,
                                             bool(self.fused_forward), self.activation_dtype, bank_lam)
            live.versions = tuple(p._version for p in params)
            self._lambda_cache = None  # lambdas_dict[last_lambda] was refreshed in place (SCENE_Net.py:333)
            return (out, act) if return_bank_activations else out
        with torch.no_grad():
            bank = self.compute_bank(x.device)
            lam = self.effective_lambdas(x.device)
            out_dtype = x.dtype if x.dtype in (torch.float32, torch.float64) else torch.float32
            if self.fused_forward and not return_bank_activations:
                if _hip.conv_fused_supported(x, ks):
                    return self.fused_served(x.contiguous(), bank, lam, self.activation_dtype or out_dtype)
                if x.dtype in (torch.float32, torch.float64):
                    # what the reference feeds is f64 {0., 1.} (ToFullDense): a device-side check routes such grids
                    # to the int8 kernels and anything else to the fp32 contraction, without a host sync
                    return _hip.forward_auto(x.contiguous(), bank, lam)[0]
            if self.serves_prepared(x):   # the per-bank work of the int8 contraction once, fused into the bank builder
                bank, prep = self.compute_bank_prepared(x.device)
                act, out = self.contract_prepared(x.contiguous(), bank, lam, prep, return_bank_activations, out_dtype)
            else:
                act, out = _hip.conv_bank(x.contiguous(), bank, lam, want_act=return_bank_activations, want_out=True,
                                          out_dtype=out_dtype)
        return (out, act) if return_bank_activations else out


class _GeneoForwardFn(torch.autograd.Function):
    """K2 + K3 forward, (sn_conv_corr, sn_geneo_bank_bwd) backward.  P [G*SN_NPARAM + G] is the gathered parameter
    vector (the gradient target); values are read from `flat`, the buffer the nn.Parameters alias, whose frozen
    coefficient sn_effective_lambdas refreshes in place."""

    @staticmethod
    def forward(ctx, x, P, flat, meta, kernel_size, want_act, fused=False, act_dtype=None, bank_lam=None):
        G = meta["G"]
        n = G * _hip.SN_NPARAM
        p = flat[:n].view(G, _hip.SN_NPARAM)
        ctx.rider_guard = None
        if bank_lam is not None:   # (SceneNet.train_rider: the same kernels' code ran in the voxelisation's first launch)
            bank, lam = bank_lam
            if isinstance(bank_lam, _RiderOutputs):
                ctx.rider_guard = (bank_lam.ring, bank_lam.k, bank_lam.gen)   # checked by backward
        else:
            bank, lam = _hip.geneo_bank_lambdas(p, meta["kinds"], kernel_size, flat[n:], meta["order"], meta["last"])
        out_dtype = x.dtype if x.dtype in (torch.float32, torch.float64) else torch.float32
        if fused and not want_act and _hip.conv_fused_supported(x, kernel_size):
            if act_dtype is not None:   # bf16 activation storage (SceneNet.activation_dtype): the linear forward writes it
                out_dtype = act_dtype
            act, out = None, _hip.conv_fused(x, bank, lam, out_dtype=out_dtype)   # forward through linearity
        elif fused and not want_act and x.dtype in (torch.float32, torch.float64):
            act, out = None, _hip.forward_auto(x, bank, lam)[0]                    # float grid, usually {0., 1.}
        else:
            act, out = _hip.conv_bank(x, bank, lam, want_act=want_act, want_out=True, out_dtype=out_dtype)
        ctx.save_for_backward(x, out, bank, P, lam, meta["kinds"])
        ctx.kernel_size = tuple(kernel_size)
        ctx.G, ctx.last = G, meta["last"]
        if act is not None:
            ctx.mark_non_differentiable(act)
        return out, act

    @staticmethod
    def backward(ctx, gout, _gact=None):
        x, out, bank, P, lam, kinds = ctx.saved_tensors
        if ctx.rider_guard is not None:
            ring, k, gen = ctx.rider_guard
            if ring["gen"][k] != gen:
                raise RuntimeError(
                    "SceneNet backward: the bank / coefficient buffers this forward was given by train_rider() have been "
                    "handed to a later forward since (they are persistent and rewritten in place by the voxelisation's "
                    "rider): more than one other train_rider() forward ran before this backward.  Run the backward first, "
                    "or call forward without bank_lam (the forward then builds its own bank).")
        G = ctx.G
        n = G * _hip.SN_NPARAM
        # bf16 activations stay bf16 (half the bytes of the two grids this pass reads; products and sums are fp32)
        gdt = torch.bfloat16 if out.dtype == torch.bfloat16 else torch.float32
        C = _hip.conv_corr(x, gout.to(gdt).contiguous(), out.to(gdt).contiguous(), ctx.kernel_size)   # [kz,kx,ky]
        # dL/dK_g = lambda_g C and dL/dlambda_g = <K_g, C> (minus the frozen coefficient's, SCENE_Net.py:331), the
        # generator Jacobians and the packing into gP: one launch
        gP = torch.empty_like(P)
        _hip.geneo_backward(P[:n].view(G, _hip.SN_NPARAM), kinds, ctx.kernel_size, bank, lam, C, ctx.last, gP)
        return None, gP, None, None, None, None, None, None, None


class SCENE_Net(SceneNet):
    """The v1 module, SCENE_Net.py:121-226: same forward, v1 generators (cylinder_kernel, cone_kernel,
    neg_sphere_kernel), lambdas initialised in [0, 0.6] (SCENE_Net.py:174-177)."""

    GENEO_CLASSES = CLASS_OF_KEY_V1
    LAMBDA_RANGE = staticmethod(lambda n: (0, 0.6))

    def __init__(self, geneo_num=None, kernel_size=None, plot=False, device=None):
        super().__init__(geneo_num, kernel_size, plot)
        self.device = device
        if device is not None:
            self.to(device)

    def get_geneo_nums(self):
        return self.sizes

    def get_dict_parameters(self):
        return dict([(n, param.data.item()) for n, param in self.named_parameters()])


class SCENENetQuantile(nn.Module):
    """SCENE_Net.py:347-415: an ensemble of v1 SCENE_Nets, one per quantile; forward stacks them on the channel axis."""

    def __init__(self, geneo_num=None, kernel_size=None, qs=torch.tensor([0.1, 0.5, 0.9]), plot=False, device=None):
        super().__init__()
        self.scnets = nn.ModuleList([SCENE_Net(geneo_num, kernel_size, plot) for _ in range(len(qs))])
        if device is not None:
            self.scnets.to(device)
        self.qs = qs
        self.device = device

    def get_num_total_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def get_dict_parameters(self):
        return dict([(n, param.data.item()) for n, param in self.named_parameters()])

    def get_cvx_coefficients(self):
        return [scnet.get_cvx_coefficients() for scnet in self.scnets]

    def get_geneo_params(self):
        return [scnet.get_geneo_params() for scnet in self.scnets]

    def forward(self, x: torch.Tensor):
        preds = [torch.squeeze(net(x), dim=1) for net in self.scnets]
        return torch.stack(preds, dim=1).to(torch.float32)  # the reference fills a float32 torch.empty


class SCENE_Net_Class(nn.Module):
    """SCENE_Net.py:421-466: thresholded v1 SCENE_Net, (gnet(x) >= tau).to(x.dtype)."""

    def __init__(self, geneo_num=None, plot=True, gnet_requires_grad=True):
        super().__init__()
        self.gnet = SCENE_Net(geneo_num, plot=False)  # the reference passes `plot` as kernel_size (SCENE_Net.py:427)
        if not gnet_requires_grad:
            for param in self.gnet.parameters():
                param.requires_grad = False
        tau_min, tau_max = 0.2, 0.6
        self.tau = nn.Parameter((tau_max - tau_min) * torch.rand(1, dtype=torch.float)[0])

    def get_threshold(self):
        return self.tau

    def get_geneo_nums(self):
        return self.gnet.sizes

    def get_cvx_coefficients(self):
        return self.gnet.lambdas_dict

    def get_geneo_params(self):
        return nn.ParameterDict(dict([(name.replace(".", "_"), p) for name, p in self.gnet.named_parameters()
                                      if "lambda" not in name]))

    def get_dict_parameters(self):
        return dict([(n, param.data.item()) for n, param in self.gnet.named_parameters()])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        pred = self.gnet(x)
        return (pred >= self.tau.to(pred.device)).to(x.dtype if x.dtype.is_floating_point else pred.dtype)
