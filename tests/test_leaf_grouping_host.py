"""Host logic of the deferred-leaf machinery (textualdegremoval_amd/engine.py: late_leaves / _leaf / _leaf_wgrad1x1 / run_late_leaves) without a GPU:
the kernel wrappers are replaced by CPU stand-ins that record what was asked of them.  Checked: leaves are queued only while a whole-network
backward collects them; 1x1 weight-gradient requests of one shape go out as ONE grouped call per shape (in first-seen order, with consecutive
table indices), the others as single calls; every gradient reaches the collector under its own prefix, in queue order; nothing is grouped when
the collector exchanges gradients (data-parallel run) or when grouping is switched off."""
import contextlib

import pytest
import torch

from textualdegremoval_amd import engine as E


class _FakeK:
    SIDE_WGRAD = False

    def __init__(self):
        self.group_calls, self.single_calls, self.joins = [], [], 0

    # ---- what engine.py calls
    def wgrad1x1_group_key(self, x, dout, Cout, Cin, gate):
        return None if Cin < 64 else (x.shape[0], Cin, Cout, x.shape[2], x.shape[3], int(gate))

    def wgrad1x1_group(self, reqs, seq=0, want_db=True):
        self.group_calls.append((seq, len(reqs), want_db, [r[0][0, 0, 0, 0].item() for r in reqs]))
        return [(torch.full((1, r[2], r[3], 1, 1), float(r[0][0, 0, 0, 0])), torch.zeros(r[2]) if want_db else None) for r in reqs]

    def conv_wgrad(self, x, dout, Cout, Cin, KH, gate=False, want_db=False, **kw):
        self.single_calls.append((Cout, Cin, float(x[0, 0, 0, 0])))
        g = torch.full((1, Cout, Cin, 1, 1), float(x[0, 0, 0, 0]))
        return (g, torch.zeros(Cout)) if want_db else g

    def side_keep(self, *t):
        return t[0] if len(t) == 1 else t

    def lane(self, i, sync=False):
        return contextlib.nullcontext()

    def on_side(self, *keep):
        return contextlib.nullcontext()

    def lanes_join(self):
        self.joins += 1

    def side_join(self):
        pass


@pytest.fixture
def fake(monkeypatch):
    k = _FakeK()
    monkeypatch.setattr(E, 'K', k)
    monkeypatch.setattr(E, 'DEFER_WGRAD', True)
    monkeypatch.setattr(E, 'GROUP_LEAVES', True)
    return k


def _queue(G, tag, Cin, Cout, value, want_db=True):
    x = torch.full((2, Cin, 4, 8), float(value))
    d = torch.zeros(2, Cout, 4, 8)
    E.set_late_prefix(f'{tag}.')
    E._leaf_wgrad1x1((x, d), (x, d, Cout, Cin, False), lambda g, db: {'w': g, 'b': db} if want_db else {'w': g}, G, want_db=want_db)


def test_requests_of_one_shape_share_one_grouped_call(fake):
    G = {}
    with E.late_leaves(G):
        _queue(G, 'a', 128, 256, 1)
        _queue(G, 'b', 64, 128, 2)          # another shape: its own group
        _queue(G, 'c', 128, 256, 3)
        _queue(G, 'd', 32, 64, 4)           # not groupable (key None): an ordinary leaf
        _queue(G, 'e', 128, 256, 5, want_db=False)      # same channels, no bias gradient: a group of its own
        E._leaf((), lambda: {'ln': torch.ones(3)}, G)
        assert G == {} and len(E._late) == 6            # nothing ran yet
        ran = []
        E.run_late_leaves(G, lambda: ran.append('main'))
    assert ran == ['main'] and fake.joins == 1 and E._late is None
    assert [(s, n, db) for s, n, db, _ in fake.group_calls] == [(0, 2, True), (1, 1, True), (2, 1, False)]
    assert fake.group_calls[0][3] == [1.0, 3.0]                         # the two 128 -> 256 requests, in queue order
    assert fake.single_calls == [(64, 32, 4.0)]
    assert set(G) == {'a.w', 'a.b', 'b.w', 'b.b', 'c.w', 'c.b', 'd.w', 'd.b', 'e.w', 'e.ln'}
    assert G['c.w'].flatten()[0].item() == 3.0 and G['e.w'].flatten()[0].item() == 5.0 and G['d.w'].flatten()[0].item() == 4.0


def test_nothing_is_deferred_or_grouped_with_a_gradient_exchange(fake):
    class Sink(dict):
        class reducer:
            collective = True
    G = Sink()
    with E.late_leaves(G):
        assert E._late is None
        _queue(G, 'a', 128, 256, 7)
        assert 'w' in G                                                 # ran at once (prefixes are the caller's business in this mode)
    assert fake.group_calls == [] and fake.single_calls == [(256, 128, 7.0)]


def test_grouping_switched_off_runs_single_launches_in_the_deferred_pass(fake, monkeypatch):
    monkeypatch.setattr(E, 'GROUP_LEAVES', False)
    G = {}
    with E.late_leaves(G):
        _queue(G, 'a', 128, 256, 1)
        _queue(G, 'b', 128, 256, 2)
        assert G == {}
        E.run_late_leaves(G, lambda: None)
    assert fake.group_calls == [] and [c[2] for c in fake.single_calls] == [1.0, 2.0]
    assert G['a.w'].flatten()[0].item() == 1.0 and G['b.w'].flatten()[0].item() == 2.0


def test_level_mode_groups_inside_a_data_parallel_backward(fake):
    """with a gradient exchange the leaves of a LEVEL are queued and run together at its end (level_end), on the current stream, and their
    gradients reach the collector right there -- before the next level starts -- so the buckets still fill in arrival order"""
    class Sink(dict):
        class reducer:
            collective = True
    G = Sink()
    with E.late_leaves(G, level_ok=True):
        assert E._late == [] and E._level_mode
        _queue(G, 'l3.a', 128, 256, 1)
        _queue(G, 'l3.b', 128, 256, 2)
        assert len(G) == 0
        E.level_end(G)
        assert set(G) == {'l3.a.w', 'l3.a.b', 'l3.b.w', 'l3.b.b'} and fake.group_calls == [(0, 2, True, [1.0, 2.0])]
        _queue(G, 'l2.a', 128, 256, 3)
        ran = []
        E.run_late_leaves(G, lambda: ran.append('main'))        # flushes the last level, then the main chain; no lane, no join
        assert ran == ['main'] and 'l2.a.w' in G and fake.joins == 0
    assert [c[0] for c in fake.group_calls] == [0, 1] and not E._level_mode and E._late is None
