"""CPU: the ordering logic of runtime.PreparedFilters (the weight-only launches a training step re-runs on a side stream: tier 0 = the
session's cached filter planes, tier 1 = the forward pass's entries, tier 2 = the reverse sweep's) against recording stand-ins for the
torch stream / event objects: which event each reader waits for, that every tier is waited for once per step, that a disabled instance
runs everything inline, and that the solver's join covers a step that used none of the buffers."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-faster-rcnn_amd"))


class FakeEvent(object):
    count = 0

    def __init__(self):
        FakeEvent.count += 1
        self.id = FakeEvent.count
        self.recorded_on = None

    def record(self, stream):
        self.recorded_on = stream.name
        stream.log.append(("record", self.id))


class FakeStream(object):
    def __init__(self, name, log):
        self.name, self.log, self.cuda_stream = name, log, hash(name) & 0xffff

    def wait_event(self, ev):
        self.log.append(("%s waits" % self.name, ev.id))

    def wait_stream(self, other):
        self.log.append(("%s waits for stream" % self.name, other.name))


@pytest.fixture
def prepared(monkeypatch):
    import torch
    from frcnn_hip import runtime
    log = []
    main = FakeStream("main", log)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: main)
    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: FakeStream("side", log))
    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    FakeEvent.count = 0
    return runtime.PreparedFilters("cuda:0"), log, main


def test_first_step_runs_inline_and_remembers_the_plan(prepared):
    p, log, _ = prepared
    ran = []
    assert p.get(("fwd", "u", "a"), lambda: ran.append("a") or "A") == "A"
    assert p.get(("wflip", "b"), lambda: ran.append("b") or "B") == "B"
    assert ran == ["a", "b"] and list(p.plan) == [("fwd", "u", "a"), ("wflip", "b")]
    assert log == []                                     # nothing waited for, no side stream yet
    p.wait_planes(), p.join()                            # no refresh has happened: both are no-ops
    assert log == []


def test_refresh_orders_the_tiers_and_every_tier_is_waited_for_once(prepared):
    p, log, main = prepared
    ran = []
    p.get(("fwd", "u", "a"), lambda: ran.append("fwd a") or "A")
    p.get(("wflip", "b"), lambda: ran.append("bwd b") or "B")
    del ran[:]
    p.weights_changed()
    p.refresh(pre=lambda: ran.append("planes"))
    assert ran == ["planes", "fwd a", "bwd b"]           # tier 0, then the forward entries, then the sweep's
    assert log[0] == ("side waits for stream", "main")   # the update itself comes first
    rec = [e for e in log if e[0] == "record"]
    assert [r[1] for r in rec] == [1, 2, 3] and all(ev.recorded_on == "side" for ev in p.events)
    del log[:]
    # the next step: planes at the first GEMM that reads them, the forward entries at the first 3x3, the sweep's at its first use
    p.wait_planes()
    p.wait_planes()
    assert log == [("main waits", 1)]
    assert p.get(("fwd", "u", "a"), lambda: pytest.fail("ready entries are not re-run")) == "A"
    assert p.get(("fwd", "u", "a"), None) == "A"
    assert log == [("main waits", 1), ("main waits", 2)]
    assert p.get(("wflip", "b"), None) == "B"
    p.join()
    assert log == [("main waits", 1), ("main waits", 2), ("main waits", 3)]


def test_a_later_tier_covers_the_earlier_ones_and_join_covers_an_idle_step(prepared):
    p, log, _ = prepared
    p.get(("fwd", "u", "a"), lambda: "A")
    p.get(("wflip", "b"), lambda: "B")
    p.weights_changed()
    p.refresh(pre=lambda: None)
    del log[:]
    assert p.get(("wflip", "b"), None) == "B"            # the sweep's buffer first: one wait for the LAST event is enough
    p.wait_planes()
    assert p.get(("fwd", "u", "a"), None) == "A"
    assert log == [("main waits", 3)]
    p.weights_changed()
    p.refresh(pre=lambda: None)
    del log[:]
    p.join()                                             # a step that touched nothing: the solver still waits before it writes the filters
    assert log == [("main waits", 3)]


def test_new_entries_and_changed_weights_run_inline(prepared):
    p, log, _ = prepared
    p.get(("fwd", "u", "a"), lambda: "A")
    p.weights_changed()
    p.refresh()
    ran = []
    assert p.get(("wflip", "new"), lambda: ran.append(1) or "N") == "N" and ran == [1]     # not in the refreshed set
    p.weights_changed()                                  # somebody changed the filters without a refresh (restore): nothing is ready
    assert p.get(("fwd", "u", "a"), lambda: ran.append(2) or "A2") == "A2" and ran == [1, 2]
    p.invalidate()
    assert p.plan == {} and p.get(("fwd", "u", "a"), lambda: "A3") == "A3"


def test_disabled_instance_runs_everything_on_the_calling_stream(prepared):
    p, log, _ = prepared
    p.enabled = False
    ran = []
    assert p.get(("fwd", "u", "a"), lambda: ran.append("a") or "A") == "A" and p.plan == {}
    p.weights_changed()
    p.refresh(pre=lambda: ran.append("planes"))
    assert ran == ["a", "planes"] and log == [] and p.stream is None
    assert p.get(("fwd", "u", "a"), lambda: ran.append("again") or "A") == "A" and ran[-1] == "again"
