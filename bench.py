#!/usr/bin/env python3
"""Headline benchmark: CoDA 1.7B masked-diffusion sampler on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

One "step" = one full ``model.diffusion_generate`` call on the per-GPU batch of the C2 workload
(BASELINE.json configs[1]: batch 32, 128-token synthetic prompts, 256 new tokens, 128 denoise
steps, block_length 32, alg=entropy, greedy, bf16) = 128 full forwards + 128 confidence/selection
passes.  Weak scaling: every rank runs its own batch of 32 (independent prompts, no per-step
communication); RCCL is used once per step to gather the completions and a small fp32 metrics
vector per rank (evaluation/lm_eval/lm_eval/evaluator.py:523-526,620-642).  Rank 0 prints ONE JSON line.

N ranks come about in either of two ways:
  * under a launcher (``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``): RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* are in the environment and this process is one rank;
  * ``python bench.py --gpus N`` on its own: this process is only a PARENT - it makes no HIP call at all (no device
    count either), starts N children (one per device, the environment above set for each), waits for them and exits with
    their status.  Profile with ``--gpus 1`` (or a pre-launched rank): under rocprofv3 the parent would be a launcher hop.
``n_gpus`` in the line is the world size the process group itself reports.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def per_step_flops(B, L, masked_rows):
    """Algorithmic FLOPs of one denoise step: forward + lm-head over consumed rows."""
    return B * L * (2_818_572_288 + 229_376 * L) + masked_rows * 622_329_856


def window_schedule(bl, steps_b):
    from coda_amd.generation_utils import fp32_schedule
    ks, m = [], float(bl)
    ratios = fp32_schedule(steps_b, 1e-3)
    for j in range(steps_b):
        k = int(m) if j == steps_b - 1 else int(torch.tensor(m) * torch.tensor(ratios[j]))
        ks.append(k)
        m -= k
    return ks


def cpu_baseline(model, budget_steps):
    """Times the CPU oracle (the reference algorithm restated in torch, oracle/ref_cpu.py) on the C1 shape
    (batch 1) for a bounded number of denoise steps on this host's cores."""
    from oracle import ref_cpu
    d = ref_cpu.Dims()
    sd = {k: v.detach().to("cpu", torch.float32) for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1001)
    ids = torch.randint(0, 151643, (1, 128), generator=g)
    with torch.no_grad():
        ref_cpu.forward_logits(sd, d, ids[:, :16])                         # warm-up
        t0 = time.perf_counter()
        ref_cpu.diffusion_generate(sd, d, ids, max_new_tokens=256, steps=128, alg="entropy", block_length=32,
                                   max_steps=budget_steps)
        dt = time.perf_counter() - t0
    s_per_step = dt / budget_steps
    return dict(value=256.0 / (128 * s_per_step), unit="generated tokens/s", cores=torch.get_num_threads(), kind="port",
                ms_per_denoise_step=s_per_step * 1e3,
                sample=f"{budget_steps} of 128 denoise steps of the C1 shape (batch 1, 128+256 tokens, block 32, "
                       f"entropy, greedy, fp32, random-init 1.7B weights), {dt:.1f} s of CPU work, extrapolated to 128 steps")


def spawn_ranks(n: int) -> int:
    """Parent of a self-launched N-rank run: no GPU call is made here (the children initialise their own device), so
    nothing is exec'ed or forked from a process that holds the GPU.  A rank that dies takes the others down instead of
    leaving them waiting in a collective."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in procs:
                        q.terminate()
            time.sleep(0.2)
    finally:
        for q in procs:
            q.kill()
    return rc


class _StubModel:
    """--stub-cpu only (tests/test_host_logic.py): stands in for the model so the launcher, the process group, the
    gathers and the JSON line can be exercised with gloo ranks on a machine without a GPU.  Its line says so."""

    class _Eng:
        timer = None
        prune_last_layer = True

    def __init__(self, mask_id):
        self.mask_id = mask_id
        self._eng = self._Eng()

    def hip_engine(self):
        return self._eng

    def diffusion_generate(self, ids, max_new_tokens, **kw):
        if os.environ.get("CODA_BENCH_TEST_FAIL_RANK") == os.environ.get("RANK", "0"):
            raise RuntimeError("stub rank asked to fail (launcher test)")
        tail = (ids.sum(1, keepdim=True) + torch.arange(max_new_tokens)[None, :]) % 1000
        return torch.cat([ids, tail], dim=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="prompts per GPU")
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--new-tokens", type=int, default=256)
    ap.add_argument("--denoise-steps", type=int, default=128)
    ap.add_argument("--block-length", type=int, default=32)
    ap.add_argument("--layers", type=int, default=28, help="debug only; anything but 28 marks the line invalid")
    ap.add_argument("--cpu-steps", type=int, default=12, help="denoise steps timed for cpu_baseline (0 = skip)")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--lmhead-materialise-from", type=int, default=None,
                    help="A/B only: candidate-row bound from which the lm-head takes the materialised route (engine default otherwise)")
    ap.add_argument("--stub-cpu", action="store_true", help="tests only: gloo ranks on the CPU around a stub generate")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the parent makes NO HIP call, not even a device count (on ROCm that initialises the runtime): a rank whose
        # device does not exist fails in its own process and takes the others down (spawn_ranks)
        sys.exit(spawn_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # CODA_BENCH_BACKEND=gloo rehearses the N>1 code path on a box with fewer GPUs than ranks (ranks then share GPUs);
    # the driver's runs use the default: RCCL, one rank per GPU
    backend = "gloo" if args.stub_cpu else os.environ.get("CODA_BENCH_BACKEND", "nccl")
    if args.stub_cpu:
        dev = torch.device("cpu")
    else:
        if backend != "nccl":
            local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from coda_amd import CoDAConfig, CoDALanguageModel
    from coda_amd.profiling import KernelTimer
    from coda_amd.dp import gather_completions, gather_metrics

    if dist is not None:
        world = dist.get_world_size()            # what the process group (RCCL) itself reports
        rank = dist.get_rank()
    cfg = CoDAConfig(num_hidden_layers=args.layers)
    if args.stub_cpu:
        model = _StubModel(cfg.mask_token_id)
    else:
        torch.manual_seed(1234)                  # same synthetic weights on every rank
        old = torch.get_default_dtype()
        torch.set_default_dtype(torch.bfloat16)
        with torch.device(dev):
            model = CoDALanguageModel(cfg)
        torch.set_default_dtype(old)
        model.eval()
        with torch.no_grad():
            model.lm_head.weight.mul_(8.0)       # "peaked" variant: avoids all-tie confidences

    B, P, G, S, bl = args.batch, args.prompt, args.new_tokens, args.denoise_steps, args.block_length
    L = P + G
    g = torch.Generator().manual_seed(1002 + rank)
    ids = torch.randint(0, 151643, (B, P), generator=g).to(dev)
    gen = dict(max_new_tokens=G, steps=S, alg="entropy", temperature=0.0, block_length=bl, mask_token_id=cfg.mask_token_id)

    eng = model.hip_engine()
    if args.lmhead_materialise_from is not None:
        eng.materialise_min_rows = args.lmhead_materialise_from

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
            sync()

    out = None
    for _ in range(args.warmup):
        out = model.diffusion_generate(ids, **gen)
        gather_completions(out, dist)
    timer = None if (args.no_kernel_timer or args.stub_cpu) else KernelTimer(["gate_up", "qkv", "o_proj", "down"], sample_every=8)
    eng.timer = timer
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = model.diffusion_generate(ids, **gen)
        gathered = gather_completions(out, dist)
    sync()
    dt_local = time.perf_counter() - t0          # this rank's own time, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    eng.timer = None
    if dist is not None:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    masks_left_local = int((out == cfg.mask_token_id).sum().item())
    # per-rank metrics vector, gathered like the completions (the reference's DP caller gathers per-rank counts and
    # metrics: evaluator.py:523-526,620-642): [generated tokens, masks left, seconds, ms per denoise step, rows gathered]
    mvec = torch.tensor([float(B * G * args.steps), float(masks_left_local), dt_local,
                         dt_local / args.steps / args.denoise_steps * 1e3, float(gathered.shape[0])], dtype=torch.float32)
    per_rank = gather_metrics(mvec.to(dev if backend == "nccl" else "cpu"), dist).cpu()
    masks_left = int(per_rank[:, 1].sum().item())
    if rank == 0:
        n_blocks = G // bl
        ks = window_schedule(bl, S // n_blocks)
        consumed, m = 0, bl
        for k in ks:
            consumed += m
            m -= k
        rows_per_gen = consumed * n_blocks * B                      # lm-head rows actually consumed per generate
        flops_gen = S * per_step_flops(B, L, 0) + rows_per_gen * 622_329_856
        # the last decoder layer's per-token tail (o_proj, MLP: 2*2048*2048 + 6*2048*6144 FLOP per token) runs on the
        # B*block_length window rows only (engine.forward_hidden(need=...)): count what is executed, not the formula
        tail_rows_skipped = B * (L - bl) if eng.prune_last_layer and bl < L else 0
        flops_exec = flops_gen - S * tail_rows_skipped * (2 * 2048 * 2048 + 6 * 2048 * 6144)
        ms_gen = dt / args.steps * 1e3
        tok_s = float(per_rank[:, 0].sum().item()) / dt        # tokens all ranks generated / slowest rank's time
        line = {
            "metric": "generated tokens/sec (node) @256 tok, 128 steps, block 32; ms per denoise step",
            "value": tok_s, "unit": "generated tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_gen, "ms_per_denoise_step": ms_gen / S, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16",
            "data": "STUB generate on the CPU (launcher rehearsal, not a measurement)" if args.stub_cpu else
                    "synthetic prompts (randint), random-init weights (lm_head x8)",
            "config": {"workload": "C2: CoDA-1.7B diffusion_generate, batch 32/GPU, prompt 128, 256 new tokens, "
                                   "steps=128, block_length=32, alg=entropy, greedy", "global_batch": world * B,
                       "seq_len": L, "parallelism": f"dp{world}", "layers": args.layers,
                       "valid": (not args.stub_cpu) and args.layers == 28 and (B, P, G, S, bl) == (32, 128, 256, 128, 32)},
            "masks_left": masks_left,
            "per_rank": {"generated_tokens": per_rank[:, 0].tolist(), "masks_left": per_rank[:, 1].tolist(),
                         "seconds": [round(v, 4) for v in per_rank[:, 2].tolist()],
                         "ms_per_denoise_step": [round(v, 3) for v in per_rank[:, 3].tolist()],
                         "rows_after_gather": per_rank[:, 4].tolist()},
            "step_roofline": {"bound": "mfma", "achieved": flops_exec / (ms_gen * 1e-3) / 1e12, "peak": PEAK_BF16_TFLOPS,
                              "unit": "TFLOP/s", "frac": flops_exec / (ms_gen * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                              "survey_formula_tflops": flops_gen / (ms_gen * 1e-3) / 1e12,
                              "note": "whole denoise step per GPU, EXECUTED FLOPs: SURVEY.md 8d's count (window mode) minus the "
                                      "last layer's o_proj/MLP on rows whose logits nobody reads; survey_formula_tflops "
                                      "uses the unreduced formula"},
        }
        if timer is not None:
            summ = timer.summary()
            line["kernels"] = summ
            dom = summ.get("gate_up")
            if dom:
                # NOT measured in this run: PMC counters need their own rocprofv3 passes (tools/prof_passes.sh,
                # tools/mfma_pmc.sh); the committed reduction of the latest such passes is carried along, with its source
                traffic = mfma_busy = traffic_src = mfma_src = None
                tpath = os.path.join(ROOT, "profiles", "traffic.json")
                if os.path.exists(tpath):
                    tj = json.load(open(tpath))
                    traffic, mfma_busy = tj.get("gate_up_hbm_bytes_per_launch"), tj.get("gate_up_mfma_util")
                    traffic_src, mfma_src = tj.get("source"), tj.get("gate_up_mfma_util_source")
                line["roofline"] = {"bound": "mfma", "kernel": "gemm256w4_bf16_kernel<SWIGLU> (gate|up projection + SiLU*up, four-wave persistent 256x256 engine)",
                                    "mfma_busy_pmc": mfma_busy,
                                    "achieved": dom["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                    "frac": dom["tflops"] / PEAK_BF16_TFLOPS, "traffic": traffic,
                                    "traffic_source": f"{traffic_src} (rocprofv3 --pmc passes of an earlier run of this command, not this run)",
                                    "mfma_busy_pmc_source": mfma_src,
                                    "avg_launch_ms": dom["avg_ms"], "flops_per_launch": dom["flops_per_launch"],
                                    "launches_timed": dom["launches"]}
        if world == 1 and args.cpu_steps > 0 and not args.stub_cpu:
            line["cpu_baseline"] = cpu_baseline(model, args.cpu_steps)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
