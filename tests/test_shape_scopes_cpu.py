"""CPU: the per-shape buffer scopes of frcnn_hip.runtime.Session (round 6; VERDICT r5 "missing 1").  The reference's graph takes any
[1, H, W, 3] image and test_net walks an imdb of hundreds of sizes (lib/nets/network.py:386-390, lib/model/test.py:138-185); here every
shape owns static buffers, so the session keeps them per shape and least-recently-used: what is tested is the bookkeeping -- which store
a buffer request lands in, what an eviction drops (buffers, the captured graph, recorded steps that address the buffers, a running
stream search), that weight-shaped buffers stay with the session, and that scratch follows the scope.  The device behaviour (memory
bounded over 64 shapes, bit-identical detections) is tests/test_streaming_shapes_gpu.py."""
import collections

import pytest
import torch

from frcnn_hip import ops
from frcnn_hip.runtime import PreparedFilters, Session, VariableStore


@pytest.fixture
def sess(monkeypatch):
    s = Session.__new__(Session)                     # (Session() needs a GPU; the bookkeeping does not)
    VariableStore.__init__(s, 3)
    s.device = torch.device("cpu")
    s.buffers, s.scopes, s.scope_group = {}, collections.OrderedDict(), {}
    s.prepared = PreparedFilters(s.device)
    s.synced = []
    monkeypatch.setattr(torch.cuda, "synchronize", lambda device=None: s.synced.append(device))
    yield s
    ops.scope_store = None


def test_requests_land_in_the_active_scope_and_the_session_otherwise(sess):
    g = sess.buf("w", (4,))
    with sess.shape_scope("A", group="t", cap=2):
        a = sess.buf("act", (8,))
        assert sess.buf("act", (8,)) is a and sess.buf("w", (4,)) is not g          # same name inside a scope: the scope's own buffer
        with ops.unscoped():
            assert sess.buf("w", (4,)) is g                                          # weight-shaped: the session's
        ws = ops.workspace(100, sess.device, "x")
        assert ops.workspace(50, sess.device, "x") is ws
        big = ops.workspace(200, sess.device, "x")                                   # grows: the old one is retired INTO the scope
        assert big is not ws and sess.scopes["A"][("ws_retired",)] == [ws]
    assert ops.scope_store is None
    assert set(sess.buffers) == {("w", (4,), torch.float32)}
    assert ("act", (8,), torch.float32) in sess.scopes["A"] and sess.scope_bytes("A") >= 8 * 4 + 4 * 4 + 200
    with sess.shape_scope("A", group="t", cap=2):
        assert sess.buf("act", (8,)) is a                                            # re-entered: the same buffers (a captured graph addresses them)
    outside = ops.workspace(10, sess.device, "x")
    assert outside is not big                                                        # outside a scope: the module-wide scratch, as before


def test_lru_eviction_drops_buffers_graphs_and_recordings_together(sess):
    class Picker(object):
        done = False
    for i, k in enumerate(["A", "B", "C"]):
        with sess.shape_scope(k, group="t", cap=3):
            sess.buf("act", (16 * (i + 1),))
        sess.graphs[k] = ("graph", k)
    sess.graphs[("train_replay", 1)] = dict(rec="r1", scope="A", picker=Picker())
    sess.graphs[("train_replay", 2)] = dict(rec="r2", scope="B")
    with sess.shape_scope("other", group="u", cap=1):                                # another group: counts separately
        sess.buf("act", (4,))
    sess.picking = True
    with sess.shape_scope("A", group="t", cap=3):                                    # touch A: B is now the least recently used of group t
        pass
    assert sess.synced == []
    with sess.shape_scope("D", group="t", cap=3):
        sess.buf("act", (4,))
    assert len(sess.synced) == 1                                                     # the device was idle before anything was freed
    assert list(sess.scopes) == ["C", "other", "A", "D"] and "B" not in sess.graphs and ("train_replay", 2) not in sess.graphs
    assert sess.picking and ("train_replay", 1) in sess.graphs                       # A's recording and its running stream search live on
    with sess.shape_scope("E", group="t", cap=2):                                    # a smaller cap drops as many as needed: C and A
        pass
    assert list(sess.scopes) == ["other", "D", "E"] and not sess.picking             # ... and the search that ran on A's recording ended with it
    assert set(sess.graphs) == set()
    assert sess.drop_scope("nope") is False and sess.drop_scope("D") is True


def test_prepared_filter_images_are_session_wide_even_when_first_built_inside_a_scope(sess, monkeypatch):
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: None)
    made = []

    def fn():
        made.append(sess.buf("bwd/wflip/conv", (3, 3)))
        return made[-1]
    with sess.shape_scope("A", group="t", cap=1):
        out = sess.prepared.get(("bwd", "wflip"), fn)
        sess.buf("act", (8,))
    with sess.shape_scope("B", group="t", cap=1):                                    # A is evicted
        pass
    assert "A" not in sess.scopes and out is sess.buffers[("bwd/wflip/conv", (3, 3), torch.float32)]
    assert fn() is out                                                               # what a later refresh() (outside any scope) writes: the same tensor
