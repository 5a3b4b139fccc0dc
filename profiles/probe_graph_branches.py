"""Do two branches of a captured hipGraph run concurrently?  Two 1-block spin kernels (torch.cuda._sleep) on two streams,
captured into one graph: ~1x the single duration = concurrent, ~2x = serialised by the graph executor."""
import torch
torch.cuda.init()
s, s2 = torch.cuda.Stream(), torch.cuda.Stream()
CYC = 2_000_000


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g1 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g1, stream=s):
        torch.cuda._sleep(CYC)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g2, stream=s):
        ev = torch.cuda.Event(); ev.record(s)
        torch.cuda._sleep(CYC)
        s2.wait_event(ev)
        with torch.cuda.stream(s2):
            torch.cuda._sleep(CYC)
            ev2 = torch.cuda.Event(); ev2.record(s2)
        s.wait_event(ev2)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g3, stream=s):
        torch.cuda._sleep(CYC); torch.cuda._sleep(CYC)


def eager2():
    ev = torch.cuda.Event(); ev.record()
    torch.cuda._sleep(CYC)
    s2.wait_event(ev)
    with torch.cuda.stream(s2):
        torch.cuda._sleep(CYC)
    torch.cuda.current_stream().wait_stream(s2)


print(f'graph, one spin: {timed(g1.replay):.3f} ms   graph, two spins in sequence: {timed(g3.replay):.3f} ms   '
      f'graph, two spins on two branches: {timed(g2.replay):.3f} ms   eager, two streams: {timed(eager2):.3f} ms')
