"""bench.py's logits rings (N > 1): the slot / ring / collective index arithmetic of `RingSchedule`, driven on the CPU
with a fake collective.  Every step's logits must reach exactly one gather, a ring must never be rewritten while its
gather is outstanding, and drain() must send a partly filled ring with only its fresh slots counted as new."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


class FakeRings:
    """Two rings of G slots; gather(k) snapshots ring k (as the asynchronous collective would read it LATER: the snapshot
    is taken when the handle is waited for, so a ring rewritten before its wait shows up as corruption)."""

    def __init__(self, G):
        self.G = G
        self.ring = [[None] * G, [None] * G]
        self.outstanding = {}
        self.delivered = []          # (ring, tuple of slot values) per completed gather
        self.nh = 0

    def gather(self, k):
        self.nh += 1
        self.outstanding[self.nh] = k
        return self.nh

    def wait(self, h):
        k = self.outstanding.pop(h)
        self.delivered.append((k, tuple(self.ring[k])))


@pytest.mark.parametrize("G", [1, 4, 16])
@pytest.mark.parametrize("steps", [1, 15, 16, 17, 33])
def test_every_step_is_gathered_once_and_no_ring_is_rewritten_early(G, steps):
    fr = FakeRings(G)
    meta, order = {}, []

    def gather(k):
        h = fr.gather(k)
        meta[h] = (k, sch.fresh[k])              # slots of the ring that are new at the time the collective starts
        return h

    def wait(h):
        fr.wait(h)
        order.append(meta[h])

    sch = bench.RingSchedule(G, gather, wait)
    for region in range(3):                      # three timed regions back to back, a drain behind each
        base = region * steps
        for i in range(steps):
            j = sch.begin_step()
            k, slot = j // G, j % G
            assert all(kk != k for kk in fr.outstanding.values()), "ring %d rewritten while its gather is outstanding" % k
            fr.ring[k][slot] = base + i
            sch.end_step(j)
        sch.drain()
        assert not fr.outstanding and sch.works == [None, None]
        assert sch.count % G == 0                # the next region starts a ring
    seen = []
    assert len(order) == len(fr.delivered)
    for (k, fresh), (kd, vals) in zip(order, fr.delivered):
        assert k == kd and 1 <= fresh <= G
        seen += list(vals[:fresh])
    assert sorted(seen) == list(range(3 * steps)), "some step's logits were gathered twice or never"


def test_drain_counts_only_the_slots_the_region_wrote():
    G = 16
    fr = FakeRings(G)
    sch = bench.RingSchedule(G, fr.gather, fr.wait)
    for i in range(17):
        j = sch.begin_step()
        fr.ring[j // G][j % G] = i
        sch.end_step(j)
    assert sch.fresh == [16, 1]
    sch.drain()
    kinds = [e[0] for e in sch.log]
    assert kinds == ["gather", "drain"] and sch.log[-1] == ("drain", 1, 1)
    # the drained ring still holds None (never written) in the other 15 slots: they are not counted as results
    assert fr.delivered[-1][1][0] == 16 and all(v is None for v in fr.delivered[-1][1][1:])
