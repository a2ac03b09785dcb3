"""CPU tests of the host side: the module contract the reference's Lightning wrapper relies on
(lit_model_wrappers.py:131-134,155,168,179), parameter packing, lambda arithmetic, sharding, and that
nothing silently falls back to a CPU path."""
import json
import os

import numpy as np
import pytest
import torch

import scene_net_amd as sna
from scene_net_amd import _hip
from scene_net_amd.geneos import pack_params
from scene_net_amd.voxelization import _size_mode_bounds
from oracle import geneo_oracle as go
from oracle import voxel_oracle as vo


@pytest.fixture(scope="module")
def contract(golden_dir):
    with open(os.path.join(golden_dir, "module_contract.json")) as f:
        return json.load(f)


def test_state_dict_and_accessor_names_match_reference(contract):
    torch.manual_seed(contract["seed"])
    m = sna.SceneNet({"cy": 2, "cone": 1, "neg": 1}, (9, 9, 9))
    assert list(m.state_dict().keys()) == contract["state_dict_keys"]
    assert [n for n, _ in m.named_parameters()] == contract["named_parameters"]
    assert list(m.get_geneo_params().keys()) == contract["geneo_param_names"]
    assert list(m.get_model_parameters_in_dict().keys()) == contract["in_dict_keys"]
    assert list(m.get_cvx_coefficients().keys()) == contract["cvx_keys"]
    for n, p in m.named_parameters():
        if "lambda" not in n:
            assert p.requires_grad == contract["requires_grad"][n], n
    assert sum(1 for n, p in m.named_parameters() if "lambda" in n and not p.requires_grad) == \
        contract["n_frozen_lambdas"]
    assert m.get_num_total_params() == contract["num_total_params"]


def test_seeded_construction_draws_the_reference_values(contract):
    torch.manual_seed(contract["seed"])
    m = sna.SceneNet({"cy": 2, "cone": 1, "neg": 1}, (9, 9, 9))
    assert m.last_lambda == contract["seeded_last_lambda"]
    mine = m.get_model_parameters_in_dict()
    for k, v in contract["seeded_values"].items():
        assert mine[k] == v, k
    # convex: sum(lambda) == 1 after construction (SCENE_Net.py:290-291)
    assert abs(sum(float(p) for p in m.lambdas_dict.values()) - 1.0) < 1e-6
    # all parameters are fp32 scalars (LitWrapperModel logs .data.item())
    for _, p in m.named_parameters():
        assert p.dtype == torch.float32 and p.dim() == 0


def test_default_geneo_num_and_kernel_size():
    torch.manual_seed(0)
    m = sna.SceneNet()
    assert list(m.geneos.keys()) == ["cy_0", "cone_0", "neg_0"]
    assert m.kernel_size_of_bank() == (9, 9, 9)
    assert not hasattr(m, "kernel_size")  # only set when given (SCENE_Net.py:256-257)
    m2 = sna.SceneNet({"cy": 1, "cone": 1, "neg": 1}, (9, 5, 5))
    assert m2.kernel_size == (9, 5, 5) and m2.kernel_size_of_bank() == (9, 5, 5)


def test_load_reference_checkpoint_values():
    # the 13 trained scalars of the committed Lightning checkpoint, keys without the `model.` prefix
    sd = {"geneos.cy_0.geneo_params.radius": 0.998896, "geneos.cy_0.geneo_params.sigma": 1.199054,
          "geneos.cone_0.geneo_params.apex": 0.0, "geneos.cone_0.geneo_params.cone_inc": 0.565547,
          "geneos.cone_0.geneo_params.cone_radius": 4.000988, "geneos.cone_0.geneo_params.radius": 1.5,
          "geneos.cone_0.geneo_params.sigma": 0.955910, "geneos.neg_0.geneo_params.neg_factor": 0.127053,
          "geneos.neg_0.geneo_params.radius": 3.000918, "geneos.neg_0.geneo_params.sigma": 0.605097,
          "lambdas_dict.lambda_cone_0": 0.608911, "lambdas_dict.lambda_cy_0": 0.024178,
          "lambdas_dict.lambda_neg_0": 0.366911}
    m = sna.SceneNet({"cy": 1, "cone": 1, "neg": 1}, (9, 5, 5))
    m.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    assert m.get_model_parameters_in_dict()["cone_0.cone_radius"] == pytest.approx(4.000988)


def test_pack_params_slots():
    p = {"radius": torch.tensor(2.0), "sigma": torch.tensor(1.5), "apex": torch.tensor(3.7),
         "cone_radius": torch.tensor(2.5), "cone_inc": torch.tensor(0.3)}
    v = pack_params(_hip.SN_GENEO_CONE, p, "cpu")
    assert v.shape == (_hip.SN_NPARAM,) and v.dtype == torch.float32
    assert v.tolist() == pytest.approx([2.0, 1.5, 3.7, 2.5, 0.3, 0.0, 0.0, 0.0])
    v = pack_params(_hip.SN_GENEO_NEG, {"radius": 3.0, "neg_factor": 0.2}, "cpu")  # sigma defaults to 1
    assert v.tolist() == pytest.approx([3.0, 1.0, 0.0, 0.0, 0.0, 0.2, 0.0, 0.0])
    with pytest.raises(KeyError):
        pack_params(_hip.SN_GENEO_CONE, {"radius": 1.0, "apex": 1.0, "cone_radius": 1.0}, "cpu")


def test_packed_params_cache_and_apex_validation():
    torch.manual_seed(1)
    m = sna.SceneNet({"cy": 1, "cone": 1, "neg": 1}, (9, 9, 9))
    a, kinds = m.packed_params(torch.device("cpu"))
    b, _ = m.packed_params(torch.device("cpu"))
    assert a is b and kinds.tolist() == [0, 1, 2]
    with torch.no_grad():
        m.geneos["cy_0"].geneo_params["radius"].fill_(3.25)
    c, _ = m.packed_params(torch.device("cpu"))
    assert c is not a and c[0, _hip.SN_P_RADIUS].item() == 3.25
    with torch.no_grad():
        m.geneos["cone_0"].geneo_params["apex"].fill_(12.0)  # > kz: reference fails in torch.stack
    with pytest.raises(RuntimeError):
        m.packed_params(torch.device("cpu"))


def test_effective_lambdas_matches_oracle_and_mutates_last():
    torch.manual_seed(5)
    m = sna.SceneNet({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9))
    names = list(m.geneos.keys())
    with torch.no_grad():
        for i, n in enumerate(names):
            m.lambdas_dict[f"lambda_{n}"].fill_(0.013 * i - 0.07)
    raw = [float(m.lambdas_dict[f"lambda_{n}"]) for n in names]
    last = names.index(m.last_lambda.replace("lambda_", ""))
    want = go.effective_lambdas(raw, last, names)
    before = m.lambdas_dict[m.last_lambda]
    got = m.effective_lambdas(torch.device("cpu"))
    assert torch.equal(got, want)
    after = m.lambdas_dict[m.last_lambda]
    assert after is not before and not after.requires_grad  # SCENE_Net.py:333
    assert abs(sum(float(p) for p in m.lambdas_dict.values()) - 1.0) < 1e-6


def test_missing_mandatory_parameter_is_keyerror():
    for cls, kw in [(sna.cylinderv2, {}), (sna.arrow, {"radius": torch.tensor(1.0)}),
                    (sna.arrow, {"radius": torch.tensor(1.0), "apex": torch.tensor(1.0)}),
                    (sna.negSpherev2, {"radius": torch.tensor(1.0)})]:
        with pytest.raises(KeyError):
            cls("g", (9, 9, 9), **kw)


def test_no_cpu_fallback_anywhere():
    torch.manual_seed(0)
    m = sna.SceneNet({"cy": 1, "cone": 1, "neg": 1}, (5, 5, 5))
    with pytest.raises(sna.HipLibraryError):
        m(torch.zeros(1, 1, 8, 8, 8, dtype=torch.float64))
    with pytest.raises(sna.HipLibraryError):
        _hip.conv_bank(torch.zeros(1, 1, 4, 4, 4), torch.zeros(1, 3, 3, 3), None, want_act=True, want_out=False)
    if not torch.cuda.is_available():
        with pytest.raises((sna.HipLibraryError, RuntimeError, AssertionError)):
            sna.hist_on_voxel(np.random.rand(10, 3))
        with pytest.raises(sna.HipLibraryError):
            sna.cylinderv2("cy", (9, 9, 9), radius=torch.tensor(2.0))
    # the product never imports the oracle
    import scene_net_amd.scene_net as a, scene_net_amd.voxelization as b, scene_net_amd.geneos as c  # noqa
    import scene_net_amd.pipeline as d, scene_net_amd.transforms as e  # noqa
    for mod in (a, b, c, d, e, _hip):
        src = open(mod.__file__).read()
        assert "import oracle" not in src and "from oracle" not in src, mod.__file__


def test_shard_range_partitions():
    for n in (0, 1, 7, 32, 256, 1000):
        for w in (1, 2, 3, 8):
            chunks = [sna.shard_range(n, r, w) for r in range(w)]
            assert chunks[0][0] == 0 and chunks[-1][1] == n
            for (a, b), (c, d) in zip(chunks, chunks[1:]):
                assert b == c
            sizes = [b - a for a, b in chunks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sna.shard_range(4, 2, 2)


def test_size_mode_bounds_follow_the_oracle():
    rng = np.random.default_rng(9)
    xyz = rng.uniform(0, 20, (400, 3)) * np.array([1.0, 0.4, 1.7]) + np.array([5.44e5, 4.634e6, 150.0])
    for sizes in [(0.5, 0.5, 0.5), (1.0, 2.0, 0.75), (3.0, 3.0, 3.0)]:
        g = vo.voxelgrid_compute(xyz, sizes=sizes)
        bbox = np.concatenate([xyz.min(0), xyz.max(0)])
        bounds, n = _size_mode_bounds(bbox, sizes)
        assert np.array_equal(bounds[:3], g["xyzmin"]) and np.array_equal(bounds[3:], g["xyzmax"])
        assert tuple(n) == tuple(int(v) for v in g["x_y_z"])


def test_transforms_signatures():
    v = sna.Voxelization([15], vxg_size=(64, 64, 64), vox_size=None)
    assert v.keep_labels == [15] and v.vxg_size == (64, 64, 64) and v.vox_size is None
    d = sna.ToFullDense(apply=(True, False))
    a, b = d((torch.tensor([0.0, 0.2, 1.0]), torch.tensor([0.0, 0.3, 0.0])))
    assert a.tolist() == [0.0, 1.0, 1.0] and b.tolist() == pytest.approx([0.0, 0.3, 0.0])
    t = sna.ToTensor()((np.ones((1, 2, 2, 2), dtype=np.float32), np.zeros((1, 2, 2, 2))))
    assert all(x.dtype == torch.float64 for x in t)


def test_pointbatch_validation():
    with pytest.raises(ValueError):
        sna.PointBatch.from_tiles([np.zeros((4, 2))], device="cuda:0")
    with pytest.raises(ValueError):
        sna.PointBatch.from_tiles([np.zeros((0, 3))], device="cuda:0")
    with pytest.raises(sna.HipLibraryError):
        sna.PointBatch.from_tiles([np.zeros((4, 3))], device="cpu")


# ------------------------------------------------------------------ criteria: host-side tables (no GPU needed)
def test_criterion_weight_tables_match_the_pinned_oracle():
    """WeightedMSE's per-bin tables (the only host arithmetic of the criteria) against oracle/loss_oracle.py, which is
    pinned bit-exact to the reference: the in-place frequency replacement chain (w_mse.py:124-126) and the fp32 weights."""
    import numpy as np
    from oracle import loss_oracle as lo
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "geneo_loss.npz"))
    for name in [str(c) for c in G["cases"]]:
        gt = torch.from_numpy(G[f"{name}|gt"])
        crit = sna.WeightedMSE(targets=gt, weighting_scheme_path=None, save_weighting_scheme=False, weight_alpha=1.5,
                               weight_epsilon=0.05)
        assert torch.equal(crit.freqs.cpu(), torch.from_numpy(G[f"{name}|est_freqs"]))
        crit.freqs = torch.from_numpy(G[f"{name}|freqs"])
        ranges = torch.from_numpy(G[f"{name}|ranges"])
        assert torch.equal(crit._bin_values(), lo.bin_value_table(crit.freqs, len(ranges)))
        assert torch.equal(crit._bin_weights(), lo.bin_weights(crit.freqs, 1.5, 0.05, len(ranges)))


def test_criteria_have_no_cpu_path():
    gt = (torch.rand(2, 1, 4, 4, 4) < 0.2).float()
    crit = sna.GENEO_Tversky_Loss(targets=gt, weighting_scheme_path=None, save_weighting_scheme=False)
    with pytest.raises(sna.HipLibraryError):
        crit(torch.rand(2, 1, 4, 4, 4), gt, {}, {})
    with pytest.raises(ValueError):
        sna.WeightedMSE(targets=None, weighting_scheme_path=None)   # the reference builds this error without raising it
    # every BinaryDiceLoss power / reduction is served (the fused kernel for p = 2, torch ops on the device otherwise),
    # none of them on the CPU; an unknown reduction raises at call time like the reference (dice_loss.py:50-51)
    for crit in (sna.BinaryDiceLoss(p=3), sna.BinaryDiceLoss(reduction="none"), sna.BinaryDiceLoss()):
        with pytest.raises(sna.HipLibraryError):
            crit(torch.rand(2, 8), gt.reshape(2, -1)[:, :8])
    with pytest.raises(Exception, match="Unexpected reduction"):
        sna.BinaryDiceLoss(reduction="median")(torch.rand(2, 8), torch.rand(2, 8))


def test_scene_net_picks_the_linear_forward_only_for_what_it_serves():
    from scene_net_amd import _hip
    assert _hip.conv_fused_supported(torch.zeros((1, 1, 8, 8, 8), dtype=torch.bool), (9, 9, 9))
    assert not _hip.conv_fused_supported(torch.zeros((1, 1, 8, 8, 8)), (9, 9, 9))                       # float grid
    assert not _hip.conv_fused_supported(torch.zeros((1, 1, 8, 8, 6), dtype=torch.bool), (9, 9, 9))     # Y % 4
    assert not _hip.conv_fused_supported(torch.zeros((1, 1, 8, 8, 8), dtype=torch.bool), (3, 3, 19))    # window > 32 bytes
    assert not _hip.conv_fused_supported(torch.zeros((1, 1, 8, 8, 8), dtype=torch.bool), (13, 13, 9))   # table > LDS
    assert sna.SceneNet.fused_forward is True


@pytest.mark.parametrize("name", ["r01_final_bench.json", "r02_bench.json", "r03_bench.json", "r04_bench.json"])
def test_committed_bench_line_keeps_the_contract(name):
    """profiles/<name> (stdout of bench.py on the MI355X box) carries every key the driver's contract names, the roofline
    and CPU-baseline objects, and internally consistent figures (round 2 adds the fp32 and cold figures and the ranks)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", name)))
    if not name.startswith("r01"):
        for k in ("value_fp32", "ms_per_step_fp32", "value_cold", "ms_per_step_cold", "rccl_ranks", "per_rank_tiles_per_s"):
            assert k in d, k
        assert d["rccl_ranks"] == d["n_gpus"] == len(d["per_rank_tiles_per_s"])
        assert d["value_fp32"] < d["value_cold"] <= 1.1 * d["value"]   # (a settled clock is not always the faster one)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "tiles/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    tiles = d["config"]["tiles_per_gpu"] * d["n_gpus"]
    assert abs(d["value"] - tiles / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] is None or r["traffic"] > 0
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["launch_ms"] * 1e-3) / 1e12) <= 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    if name.startswith("r04"):   # round 4: the sustained loop is long enough for an outside sampler and carries the device's own
        busy = d["sustained"]["gpu_busy_percent"]   # busy counter; the voxel stage is the one-pass form; the reference's defaults
        assert d["sustained"]["wall_s"] >= 5.0 and busy["samples"] >= 20 and busy["frac_samples_busy_ge_90"] >= 0.9
        assert "one pass" in d["roofline_voxel"]["kernel"] and d["roofline_voxel"]["traffic"] < 1.6 * d["roofline_voxel"]["bytes_per_stage"]
        rd = d["reference_defaults"]
        assert rd["head_only"]["ms_per_call"] > 0 and "K3L" in rd["head_only"]["kernel"] and rd["bound"].startswith("hbm")
    if name.startswith(("r03", "r04")):   # round 3: the sustained figure, and the dominant kernel is the z-walk
        assert d["value_sustained"] >= d["value"] and "conv_occ_i8z_kernel" in r["kernel"] and r["launches_timed"] >= 3
        assert abs(d["sustained"]["value"] - tiles / (d["sustained"]["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


def test_device_mismatch_is_refused_before_any_launch():
    """a C-ABI call takes raw pointers and one stream; operands on different devices must be refused."""
    from types import SimpleNamespace as NS
    import scene_net_amd as sna
    a = NS(is_cuda=True, device=torch.device("cuda:0"))
    b = NS(is_cuda=True, device=torch.device("cuda:1"))
    assert sna._hip._common_device([a, None, a]) == torch.device("cuda:0")
    assert sna._hip._common_device([None, torch.zeros(1)]) is None   # CPU tensors: _ptr's own message
    with pytest.raises(sna.HipLibraryError, match="different devices"):
        sna._hip._common_device([a, b])
    # every tensor-taking wrapper is guarded
    for name in ("conv_bank", "conv_fused", "forward_auto", "geneo_bank", "voxel_occupancy_fused", "voxel_scatter",
                 "conv_corr", "loss_forward", "loss_backward", "gather_points"):
        assert hasattr(getattr(sna._hip, name), "__wrapped__"), name


def test_voxelization_in_a_badly_forked_worker_says_what_to_do(monkeypatch):
    import scene_net_amd as sna
    monkeypatch.setattr(torch.cuda, "_is_in_bad_fork", lambda: True)
    with pytest.raises(sna.HipLibraryError, match="spawn"):
        sna.Voxelization([15], vxg_size=(8, 8, 8))((np.zeros((4, 3)), np.zeros(4)))


def test_prepared_verdict_is_copy_and_pickle_safe():
    """_hip.PreparedVerdict lives in a module's __dict__ (contract_prepared / fused_served): copying or pickling the module
    must neither fail on its event / pinned buffer nor carry a learnt verdict over to the copy."""
    import copy
    import pickle
    from scene_net_amd import _hip
    v = _hip.PreparedVerdict()
    v._key, v._state = ("k",), 2
    for w in (copy.deepcopy(v), pickle.loads(pickle.dumps(v))):
        assert isinstance(w, _hip.PreparedVerdict) and w._key is None and w._state == 0


def test_round3_entry_points_are_bound():
    """every entry point added in round 3 is declared in the header, exported by the library and bound in _hip.SYMBOLS"""
    from scene_net_amd import _hip
    lib = _hip.load()
    for name in ("sn_voxel_occupancy_fused_bank", "sn_voxel_occupancy_sized_bank", "sn_conv_corr_ws", "sn_conv_corr_ws_bytes",
                 "sn_conv_fused_v", "sn_conv_fused_prep", "sn_conv_fused_prep_bytes", "sn_conv_fused_prepared",
                 "sn_loss_forward_m", "sn_loss_backward_u", "sn_criterion_forward", "sn_criterion_backward",
                 "sn_geneo_bank_prep", "sn_conv_bank_prep", "sn_conv_bank_prepared", "sn_conv_bank_prepared_served"):
        assert name in _hip.SYMBOLS and hasattr(lib, name), name
    assert lib.sn_version() >= 102
    assert int(lib.sn_conv_fused_prep_bytes(9, 9, 9)) > 0 and int(lib.sn_conv_fused_prep_bytes(9, 9, 40)) == 0
    assert int(lib.sn_conv_corr_ws_bytes(_hip.SN_OCC8, 32, 64, 64, 64, 9, 9, 9)) > \
        int(lib.sn_conv_corr_ws_bytes(_hip.SN_F32, 32, 64, 64, 64, 9, 9, 9)) > 0


def test_tracked_parameters_stay_ordinary_parameters():
    """The model's parameters carry a Parameter subclass that counts `.data` accesses (scene_net._TrackedParameter: the caches
    see a write through .data).  It must change nothing a user of nn.Parameter can observe: isinstance, repr, state_dict keys
    and values, deepcopy (keeps the subclass), pickle (comes back a plain Parameter and is re-tagged lazily), optimisers on
    torch's multi-tensor path; and `.data` reads / writes / assignments bump the touch counter the cache keys carry."""
    import copy
    import io
    from scene_net_amd import scene_net as sn
    m = sna.SceneNet({"cy": 2, "cone": 1, "neg": 1}, (9, 9, 9))
    plain_repr = {n: repr(p) for n, p in m.named_parameters()}
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    for p in m.parameters():
        sn._track(p)
    assert all(type(p) is sn._TrackedParameter and isinstance(p, torch.nn.Parameter) for p in m.parameters())
    assert {n: repr(p) for n, p in m.named_parameters()} == plain_repr
    assert list(m.state_dict()) == list(sd0) and all(torch.equal(m.state_dict()[k], sd0[k]) for k in sd0)
    assert all(type(p) is sn._TrackedParameter for p in copy.deepcopy(m).parameters())
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    assert all(isinstance(p, torch.nn.Parameter) for p in m2.parameters())
    m2.load_state_dict(m.state_dict())
    from torch.optim import optimizer as topt
    assert sn._TrackedParameter in topt._foreach_supported_types
    p = next(q for q in m.parameters() if q.requires_grad)
    t0, v0 = sn._DATA_TOUCHES[0], p._version
    p.data.add_(0.5)
    assert sn._DATA_TOUCHES[0] == t0 + 1 and p._version == v0          # invisible to _version, counted here
    p.data = p.data.clone()
    assert sn._DATA_TOUCHES[0] >= t0 + 3
    t1 = sn._DATA_TOUCHES[0]
    with torch.no_grad():
        p.add_(1.0)                                                     # an ordinary in-place op: version, not the counter
    assert sn._DATA_TOUCHES[0] == t1 and p._version == v0 + 1
    opt = torch.optim.SGD([q for q in m.parameters() if q.requires_grad], lr=0.1, momentum=0.9)
    for q in m.parameters():
        if q.requires_grad:
            q.grad = torch.ones_like(q)
    opt.step()
    assert sn._DATA_TOUCHES[0] == t1                                    # optimiser steps do not go through .data
