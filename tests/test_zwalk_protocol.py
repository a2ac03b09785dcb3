"""The z-walk kernel's hand-over protocol (csrc/conv_i8z.inc; SceneNet.forward's contraction, core/models/SCENE_Net.py:322-339)
checked WITHOUT a GPU, on the two levels a device run cannot prove anything about:

* the protocol itself -- tools/debug/zwalk_protocol_sim.py: a host model that keeps the ground truth beside the counters
  (which plane every ring row holds, which LDS-DMA is in flight, which rounds are reading) and an adversarial scheduler;
  it must find no hazard in the protocol the kernel carries, and it must FIND the read[] hazard of the round-3 protocol
  (so that "no hazard" means something);
* the instruction order hipcc emits around every counter access -- tools/zwalk_handover_audit.py on the cross-compiled ISA
  of every instantiation; it must hold, and it must notice when the ISA is tampered with.

(The third level, the hardware's ordering of LDS-DMA / DS operations against an LDS counter, was measured on the part:
tools/micro/ldsdma_handover.hip, profiles/r04_ldsdma_handover.txt.)"""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


sim = _load(os.path.join(ROOT, "tools", "debug", "zwalk_protocol_sim.py"), "zwalk_protocol_sim")
audit = _load(os.path.join(ROOT, "tools", "zwalk_handover_audit.py"), "zwalk_handover_audit")


# ------------------------------------------------------------------------------------------------ the protocol model
@pytest.mark.parametrize("kTPS,nwaves", [(4, 12), (8, 12), (8, 8)])     # the three shapes sn_set_option("conv_i8z_variant") offers
@pytest.mark.parametrize("my_jobs,LZ", [(1, 1), (1, 8), (1, 64), (2, 26), (3, 5), (5, 1), (1, 128)])
def test_shipped_protocol_has_no_hazard(kTPS, nwaves, my_jobs, LZ):
    for skip in (0.0, 0.3):                 # (rounds skipped: partial columns, a last z segment shorter than the others)
        for adversarial in (False, True):
            for seed in range(3):
                assert sim.simulate(my_jobs, LZ, kTPS, nwaves, seed, ordered_reads=True, skip_prob=skip, adversarial=adversarial)


@pytest.mark.parametrize("kTPS,nwaves", [(4, 12), (8, 12), (2, 12), (4, 16)])
def test_directed_adversary_finds_nothing_with_ordered_reads(kTPS, nwaves):
    """a round frozen at each of the last slots of a stream while everything else runs as far as it is let"""
    for LZ in (8, 20, 64):
        NPL = LZ + 8
        for fs in range(max(0, NPL - 13), NPL + 3):
            for seed in range(2):
                assert sim.simulate(1, LZ, kTPS, nwaves, seed, ordered_reads=True, adversarial=False, freeze_slot=fs)


def test_the_model_finds_the_read_counter_hazard_of_the_round_3_protocol():
    """Without the ordered-reads role a ticket of the flush slots completes read[v & 15] while a round of slot v = NPL - 6 is
    still reading: the fold of plane NPL - 1 overwrites plane NPL - 17 under it.  Reachable with two tickets per slot (the
    retired four-rounds shape) and with 16 waves (the retired 16-wave shape) ..."""
    for kTPS, nwaves in ((2, 12), (4, 16)):
        with pytest.raises(sim.Hazard, match="was overwritten by plane"):
            sim.simulate(1, 64, kTPS, nwaves, 0, ordered_reads=False, adversarial=False, freeze_slot=64 + 8 - 6)
    # ... and not with the shipped 12 waves x 4 (or 8) tickets per slot: there every other wave is held by a ticket that waits
    # for slot v (the folds of planes v + 5 .., the frozen round's own LDS-DMA report) before one can reach slot v + 16
    for kTPS in (4, 8):
        for seed in range(4):
            assert sim.simulate(1, 64, kTPS, 12, seed, ordered_reads=False, adversarial=False, freeze_slot=64 + 8 - 6)


def test_the_model_notices_a_missing_dependency():
    """sanity of the model: with the LDS-DMA distance larger than the raw ring allows, it reports the overwrite"""
    with pytest.raises((sim.Hazard, AssertionError)):
        sim.simulate(1, 64, 4, 12, 0, lag=3, dd=6)


# --------------------------------------------------------------------------------------------------- the ISA audit
@pytest.fixture(scope="module")
def isa_kernels():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    if not audit.fresh():
        audit.build()            # cross-compiles conv_i8s.hip for gfx950 (no GPU needed), ~20 s
    ks = audit.kernels(open(audit.ASM).read().split("\n"))
    assert len(ks) == 12, [k for k, _ in ks]
    return ks


def test_every_instantiation_keeps_every_hand_over_rule(isa_kernels):
    for name, body in isa_kernels:
        assert audit.audit(name, body) == [], name


def _headline(isa_kernels):
    return next((n, list(b)) for n, b in isa_kernels if "IfLi1ELi2ELi12ELb0E" in n)


def test_audit_notices_a_ring_read_hoisted_above_the_spin(isa_kernels):
    name, body = _headline(isa_kernels)
    spin = next(i for i, l in enumerate(body) if "@zw:spin_exit" in l)
    ring = next(i for i in range(spin, len(body)) if body[i].strip().startswith("ds_read_b128") and audit.is_ring_access(body[i]))
    line = body.pop(ring)
    claim = [i for i, l in enumerate(body) if l.strip().startswith("ds_add_rtn_u32")][1]
    body.insert(claim + 1, line)
    assert any(e.startswith("R1") for e in audit.audit(name, body))


def test_audit_notices_a_landed_report_without_its_wait(isa_kernels):
    name, body = _headline(isa_kernels)
    for i, l in enumerate(body):
        if "@zw:wait1" in l:
            body[i] = "\t; (removed)"
            j = next(k for k in range(i, i + 3) if body[k].strip().startswith("s_waitcnt"))
            body[j] = "\ts_nop 0"
    assert any(e.startswith("R3") for e in audit.audit(name, body))


def test_audit_notices_the_folded_report_above_the_folds_stores(isa_kernels):
    name, body = _headline(isa_kernels)
    a = next(i for i, l in enumerate(body) if "@zw:add2" in l)
    w = max(i for i in range(a) if body[i].strip().startswith("ds_write_b128"))
    line = body.pop(w)
    body.insert(a + 3, line)       # the store now sits below the add
    errs = audit.audit(name, body)
    assert any(e.startswith("R4") for e in errs), errs


def test_audit_notices_a_read_report_with_a_ring_read_behind_its_wait(isa_kernels):
    name, body = _headline(isa_kernels)
    w = next(i for i, l in enumerate(body) if "@zw:wait3" in l)
    j = next(k for k in range(w, w + 3) if body[k].strip().startswith("s_waitcnt"))
    body.insert(j + 1, "\tds_read_b128 v[0:3], v4 offset:256")
    assert any(e.startswith("R5") for e in audit.audit(name, body))


def test_audit_notices_a_flat_counter_read_and_a_bare_lds_dma(isa_kernels):
    name, body = _headline(isa_kernels)
    r = next(i for i, l in enumerate(body) if re.search(r"ds_read_b32 v\d+, v\d+ offset:16768", l))
    flat = list(body)
    flat[r] = "\tflat_load_dword v2, v[0:1] sc0 sc1"
    assert any(e.startswith(("R2", "R7")) for e in audit.audit(name, flat))
    bare = list(body)
    d = next(i for i, l in enumerate(bare) if "@zw:dma" in l)
    k = next(i for i in range(d, d + 8) if bare[i].strip().startswith("s_mov_b32 m0"))
    bare[k] = "\ts_nop 0"
    assert any(e.startswith("R6") for e in audit.audit(name, bare))
