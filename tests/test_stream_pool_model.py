"""CPU: the host logic of ngp_harness.streams.ensure_pool against a MODEL of how ROCm 7.2 hands out hardware queues on MI355X.

The model is what round 6 read from the runtime's own log and timed (tools/probes/hw_queue_log.py, tools/hw_queue_order_probe.py):
  * a stream gets its queue at first use; at most 4 queues per priority; once they are taken a new stream shares the queue with the fewest users,
    ties going to the most recently created queue;
  * queues are numbered in order of creation across priorities, and two DIFFERENT queues whose numbers differ by a multiple of 4 hand over slowly
    (59 us per round trip against 31 us); two streams on ONE queue run one after the other.
ensure_pool only sees the two measurements (`_handover_us`, `_run_beside`); here they are answered by the model.  Checked against what the GPU runs
recorded (`pool_report` of the `P` rows of profiles/r06_hw_queue_order.json and of tests/test_gpu_streams.py's orders): WHICH candidate became the side
stream -- the model agrees in every recorded order -- and, where the model agrees there too (4 of 7 orders: the device sets aside more range candidates
than this model predicts once queues are shared, so the model is a lower bound there), how many range candidates were tried.  In every order the
streams the pool ends up with must be independent under the model's own rules.  tests/test_gpu_streams.py asks the real device the same questions."""
import pytest
import torch

from ngp_harness import streams


class Runtime:
    def __init__(self):
        self.queues = []  # one dict per hardware queue, in order of creation: {"prio", "users"}

    def first_use(self, s):
        if s.queue is not None:
            return
        mine = [i for i, q in enumerate(self.queues) if q["prio"] == s.priority]
        if len(mine) < 4:
            self.queues.append({"prio": s.priority, "users": 0})
            s.queue = len(self.queues) - 1
        else:
            fewest = min(self.queues[i]["users"] for i in mine)
            s.queue = max(i for i in mine if self.queues[i]["users"] == fewest)
        self.queues[s.queue]["users"] += 1


class FakeStream:
    rt = None

    def __init__(self, device=None, priority=0):
        self.priority, self.queue = priority, None


@pytest.fixture
def model(monkeypatch):
    rt = FakeStream.rt = Runtime()
    null = FakeStream()
    monkeypatch.setattr(streams, "_SIDE", {})
    monkeypatch.setattr(streams, "_PARTS", {})
    monkeypatch.setattr(streams, "_REPORT", {})
    monkeypatch.setattr(streams, "_KEEP", {})
    monkeypatch.setattr(torch.cuda, "Stream", FakeStream)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: null)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(streams, "_touch", lambda s, idx: rt.first_use(s))
    monkeypatch.setattr(streams, "_handover_us", lambda a, b, idx, **kw: 59.0 if (a.queue != b.queue and (a.queue - b.queue) % 4 == 0) else 31.0)
    monkeypatch.setattr(streams, "_run_beside", lambda main, a, b, idx, **kw: a.queue != b.queue)

    def run(own):
        rt.first_use(null)
        for tok in [t for t in own.split(",") if t]:
            rt.first_use(FakeStream(priority=-1 if tok == "X" else 0))
        streams.ensure_pool("cuda:0")
        return streams.pool_report("cuda:0"), streams._SIDE[(0, -1)], streams._PARTS[0], null

    return run


# own streams used before the pool -> (side stream = candidate, range candidates tried): what the GPU recorded for the same orders
RECORDED = {"": (0, 3), "x,x,x": (1, 3), "x,X,x": (1, 4), "x,x,x,x,x": (1, 4), "x": (0, 7), "x,x": (0, 7), "X,x": (0, 7)}
MODEL_AGREES_ON_TRIED = {"", "x,x,x", "x,X,x", "x"}


@pytest.mark.parametrize("own", list(RECORDED))
def test_pool_outcome_matches_what_the_gpu_recorded(model, own):
    rep, side, parts, null = model(own)
    assert rep["side_stream_is_candidate"] == RECORDED[own][0], rep
    assert rep["range_stream_candidates_tried"] == RECORDED[own][1] if own in MODEL_AGREES_ON_TRIED else rep["range_stream_candidates_tried"] <= RECORDED[own][1], rep
    assert rep["independent_range_streams"] == 3 and len(parts) == 3
    assert side.queue != null.queue and (side.queue - null.queue) % 4 != 0  # never on the caller's queue, never 4 apart from it
    qs = [p.queue for p in parts]
    assert len(set(qs)) == 3 and len({q % 4 for q in qs}) == 3, qs  # three queues, no two of them 4 apart


def test_second_call_is_a_no_op_and_returns_the_same_streams(model):
    rep, side, parts, _ = model("")
    parts = list(parts)
    n_queues = len(FakeStream.rt.queues)
    streams.ensure_pool("cuda:0")
    assert streams.side_stream("cuda:0") is side and streams.part_streams("cuda:0", 3) == parts and len(FakeStream.rt.queues) == n_queues
    more = streams.part_streams("cuda:0", 5)  # beyond the pool: created late, whatever queue they get
    assert more[:3] == parts and len(more) == 5


def test_inside_a_capture_streams_are_created_but_nothing_is_run(model, monkeypatch):
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
    FakeStream.rt.first_use(torch.cuda.current_stream())
    streams.ensure_pool("cuda:0")
    assert streams._SIDE[(0, -1)].queue is None and all(p.queue is None for p in streams._PARTS[0]) and len(streams._PARTS[0]) == 3
    assert "capture" in streams.pool_report("cuda:0")["created"]
