"""(ON THE GPU BOX) what the R2D1 sampling step's row-commit launch costs and why: its five entries
(action int64 [T+1,B], q f32 [T,B,6], prev_rnn_state h / c f32 [T,B,1,512], the action copy for the host) with
and without the wait-reset mask, the two 128 KB entries alone, and an empty launch -- hipGraph replays of
200 launches, HIP events."""
import torch
from rlpyt_amd import ops

dev = torch.device("cuda")
T, B, lo, n = 40, 192, 64, 64
t_dev = torch.tensor([3], dtype=torch.int64, device=dev)
action_rows = torch.zeros(T + 1, B, dtype=torch.int64, device=dev)
q_rows = torch.zeros(T, B, 6, device=dev)
h_rows, c_rows = torch.zeros(T, B, 1, 512, device=dev), torch.zeros(T, B, 1, 512, device=dev)
action, q = torch.ones(n, dtype=torch.int64, device=dev), torch.ones(n, 6, device=dev)
h, c = torch.ones(n, 1, 512, device=dev), torch.ones(n, 1, 512, device=dev)
action_out = torch.zeros(n, dtype=torch.int64, device=dev)
done = torch.zeros(n, dtype=torch.uint8, device=dev)
done[5] = 1


def timed(entries, label, reps=200):
    rc = ops.RowCommit(len(entries), dev)
    rc.set_entries(entries)
    rc.launch(t_dev)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            rc.launch(t_dev)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{label:58s} {e0.elapsed_time(e1) * 1e3 / (5 * reps):7.2f} us per launch")


for zw in (None, done):
    tag = "with wait-reset mask" if zw is not None else "plain"
    full = [(action_rows, action, lo, 1, zw), (q_rows, q, lo, 0, zw), (h_rows, h, lo, 0, zw),
            (c_rows, c, lo, 0, zw), (action_out, action, None, 0, zw)]
    timed(full, f"five entries of the R2D1 step, {tag}")
    timed(full[2:4], f"h + c (2 x 128 KB), {tag}")
    timed(full[:2] + full[4:], f"action + q + host copy (small entries), {tag}")
timed([(action_out, action, None, 0)], "one 512-byte entry (an almost empty launch)")
