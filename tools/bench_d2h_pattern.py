"""Device-to-host bandwidth of the copy pattern of llsm_analyze_batch: T host threads, each on its own stream, pulling blocks
of parameter rows (32 utterances x 200 frames: eleven arrays of 25 KB ... 6.5 MB, 19 MB in all) into page-locked memory --
as eleven copies per block (what the library does) and as ONE copy of the same bytes.  torch is used for streams and memory only.
    python tools/bench_d2h_pattern.py [threads] [blocks per thread]"""
import sys
import threading
import time

import torch

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
F = 32 * 200
sizes = [F, F, F * 100, F * 100, F * 256, F * 256, F, F * 4, F, F * 16, F * 16]          # floats per array
total = sum(sizes)
dev = torch.device("cuda", 0)


def worker(split, out):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        if split:
            d = [torch.empty(n, dtype=torch.float32, device=dev) for n in sizes]
            h = [torch.empty(n, dtype=torch.float32).pin_memory() for n in sizes]
        else:
            d = [torch.empty(total, dtype=torch.float32, device=dev)]
            h = [torch.empty(total, dtype=torch.float32).pin_memory()]
        st.synchronize()
        out.append(None)
        barrier.wait()
        for _ in range(B):
            for a, b in zip(h, d):
                a.copy_(b, non_blocking=True)
            st.synchronize()


for split in (1, 0, 1, 0):
    barrier = threading.Barrier(T + 1)
    outs = []
    th = [threading.Thread(target=worker, args=(split, outs)) for _ in range(T)]
    [t.start() for t in th]
    barrier.wait(); t0 = time.perf_counter()
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    print("%d threads, %s: %.1f GB/s (%.1f ms for %d blocks of %.1f MB)" % (
        T, "11 copies per block" if split else "1 copy per block", T * B * total * 4 / dt / 1e9, dt * 1e3, T * B, total * 4 / 1e6), flush=True)
