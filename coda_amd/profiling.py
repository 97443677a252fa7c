"""Per-kernel HIP-event timing on the launch stream (used by bench.py for the roofline figure)."""
import torch


class KernelTimer:
    """Brackets every ``sample_every``-th launch of the kernels named in ``names`` with events
    recorded on the current (= launch) stream, and accumulates duration and algorithmic flops."""

    def __init__(self, names, sample_every=1):
        self.names = set(names)
        self.sample_every = sample_every
        self.seen = {}
        self.pending = {}
        self.pairs = {n: [] for n in self.names}
        self.enabled = True

    def begin(self, name, flops):
        if not self.enabled or name not in self.names:
            return
        c = self.seen.get(name, 0)
        self.seen[name] = c + 1
        if c % self.sample_every:
            return
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self.pending[name] = (e0, flops)

    def end(self, name):
        p = self.pending.pop(name, None)
        if p is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.pairs[name].append((p[0], e1, p[1]))

    def summary(self):
        """-> {name: dict(launches, avg_ms, tflops)}; call after a device synchronize."""
        out = {}
        for n, prs in self.pairs.items():
            if not prs:
                continue
            ms = [a.elapsed_time(b) for a, b, _ in prs]
            fl = sum(f for _, _, f in prs)
            out[n] = dict(launches=len(prs), avg_ms=sum(ms) / len(ms), tflops=fl / (sum(ms) * 1e-3) / 1e12,
                          flops_per_launch=fl / len(prs))
        return out
