"""Teacher-forced parity analysis shared by the GPU tests and smoke().

The HIP path and the oracle are compared step by step FROM THE SAME STATE (the oracle is restarted
from the HIP path's x at every step), so one near-tie cannot cascade.  Per step:

  tokens       x0 of every candidate equals the oracle's, except where the oracle's own winner /
               runner-up distance is inside the declared near-tie band.  The band is DERIVED from the measured
               deviation of the two bf16 forwards (FWD_ERR_ULPS_*, below): greedy - the distance of the row's two
               largest logits is at most 2 * FWD_ERR_ULPS bf16 ulps of the row's largest |logit|; sampling - the
               relative gap of the two best race keys p/q at most 1 - e^-u, u = the same 2 * FWD_ERR_ULPS ulps of the
               row's largest PRE-division logit, amplified by 1/T, plus two ulps of the bf16 quotient (logit_band(),
               token_band()); with top-p also where either path's token sits on the nucleus boundary (the mass in
               front of it crosses top_p under the same logit band).  The share of excused tokens is bounded
               (assert_parity): a band that excuses too much hides bugs.
  confidence   for candidates whose token agrees, |conf_hip - conf_oracle| <= conf_tolerance(...):
               what the same logit band does to the confidence - worst case over all probability ratios, or, tighter,
               first order through the confidence's own gradient (ref_cpu.sample_tokens 'sens') - plus the oracle's
               own bf16 rounding of the confidence.
  selection    given the HIP path's OWN confidences, the revealed positions must be exactly the
               reference rule (top-k, or top-k of softmax(conf/alg_temp)/noise; ties -> lowest index)
               and k must equal the reference's fp32 schedule.  (L0: integer logic is exact.)
Anything else is a violation."""
import contextlib
import math

import torch
import torch.nn.functional as F

from oracle import ref_cpu

# ---------------------------------------------------------------------------------- where the bands come from
# Two bf16 implementations of the same forward do not produce the same bf16 logits: every rounding point sits on another
# fp32 sum.  How far apart they land is MEASURED, per logit, in bf16 ulps of the row's largest |logit|:
#   * 28 layers, V = 151 936 (tests/test_hip_fullsize.py::test_c2_full_size_forward_logits_vs_cpu_oracle, 1152 rows x 151 936
#     logits, profiles/r03_forward_noise_floor.txt): HIP vs the bf16 CPU oracle mean 0.62, p99.9 4.5, max 9.75; among the
#     logits within 16 ulps of their row's maximum (the ones a token decision can involve) p99.9 4.0, max 7.0.  The bf16
#     oracle's own distance from fp32 arithmetic is the same: mean 0.61, p99.9 4.6, max 11.2 (HIP vs fp32: 0.61 / 4.6 / 9.9),
#     and its argmax differs from the fp32 argmax on 5.6 % of the rows of this random-init model (HIP vs bf16 oracle: 5.0 %).
#     So the deviation is the reference's own bf16 noise floor, not an error of the port, and no band can be narrower.
#   * the 2- and 3-layer test models (tests/test_hip_e2e.py::test_forward_logits_vs_oracle): p99.9 <= 1.5.
# FWD_ERR_ULPS is the p99.9 figure (rounded up); the tests that measure it assert it.  Every band below is stated in it:
# a token decision races two logits, each of which may move by FWD_ERR_ULPS ulps -> the greedy band is 2 * FWD_ERR_ULPS
# ulps of the row's largest |logit| on the distance of the row's two largest logits, and u = logit_band() is the same
# quantity for sampling and for the confidences.
FWD_ERR_ULPS_SMALL = 1.5
FWD_ERR_ULPS_FULL = 4.0
BAND_TOK = 2 * FWD_ERR_ULPS_SMALL * 2.0 ** -7   # `origin` branch only: relative form of the greedy band at the bottom of a binade


@contextlib.contextmanager
def lm_head_gain(m, gain):
    """The GPU test session's 1.7B model carries lm_head x 8 (bench.py's "peaked" weights); x 1 is a trained checkpoint's
    logit range (|logit| < 8).  Scaling by a power of two is exact in bf16 and is undone on exit; the engine reads the same
    storage, so the change is seen by the kernels at once."""
    if gain != 8.0:
        with torch.no_grad():
            m.lm_head.weight.mul_(gain / 8.0)
    try:
        yield
    finally:
        if gain != 8.0:
            with torch.no_grad():
                m.lm_head.weight.mul_(8.0 / gain)


def bf16_ulp(x: torch.Tensor) -> torch.Tensor:
    return torch.exp2(torch.floor(torch.log2(x.abs().clamp_min(1e-30))) - 7)


def logit_band(absmax: torch.Tensor, temperature: float, e: float = FWD_ERR_ULPS_SMALL) -> torch.Tensor:
    """u = bound on how far a DIFFERENCE of two softmax inputs of a row may sit from the oracle's: every bf16 logit of
    the HIP forward may differ from the oracle's by ``e`` bf16 ulps of the row's largest logit (2 e ulps on a difference;
    e = FWD_ERR_ULPS_*, measured - see the top of this file).
    ``absmax`` is the oracle's largest |softmax input|, i.e. AFTER its division by the temperature (ref_cpu.sample_tokens
    diag), so the logit that carried the ulp had magnitude absmax * T: its ulp is amplified by 1 / T once, and the bf16
    quotient is rounded again at its own magnitude."""
    if temperature <= 0:
        return bf16_ulp(absmax) * 2.0 * e
    return bf16_ulp(absmax * temperature) * 2.0 * e / temperature + bf16_ulp(absmax) * 2.0


def token_band(absmax: torch.Tensor, temperature: float, e: float = FWD_ERR_ULPS_SMALL) -> torch.Tensor:
    """Largest oracle-side decision margin at which the HIP path may pick the runner-up instead.  Greedy: the margin is
    the distance l1 - l2 of the row's two largest logits IN LOGIT UNITS and the band 2 e bf16 ulps of the row's largest
    |logit|.  Sampling: the margin is the relative gap 1 - key2 / key1 of the two best race keys p / q; moving the two logits
    by u in total scales key2 / key1 by at most e^u, so the order can flip only if 1 - gap >= e^-u, i.e. gap <= 1 - e^-u
    (always < 1: a gap of 1 is never excused)."""
    if temperature <= 0:
        return logit_band(absmax, 0.0, e)
    return -torch.expm1(-logit_band(absmax, temperature, e))


def conf_tolerance(conf_o: torch.Tensor, absmax: torch.Tensor, pmax: torch.Tensor, alg: str, temperature: float,
                   e: float = FWD_ERR_ULPS_SMALL, sens=None) -> torch.Tensor:
    """|conf_hip - conf_oracle| bound, the smaller of two:
      worst  moving every softmax input by at most u / 2 scales every probability by a factor in [e^-u, e^u] (logit_band);
             entropy: |d sum p log p| <= (e^u - 1) * (|H| + 1) to first order in the probability ratios;
      tight  first order in the inputs: |d conf| <= (u / 2) * sum_v |d conf / d l_v| (``sens``, from the oracle's own
             probabilities: ref_cpu.sample_tokens diag), with (e^u - 1) / u for the curvature.  On flat rows (|H| ~ 11) this
             is 10 x below `worst`; not used with top-p / top-k (membership of the kept set is not differentiable).
    Both plus the oracle's own bf16 rounding of its probabilities and of the confidence (2^-7 relative)."""
    u = logit_band(absmax, temperature, e)
    grow = torch.expm1(u)
    if alg == "entropy":
        base = conf_o.abs() + 1.0
    elif alg == "topk_margin":
        base = 2.0 * pmax                         # p1 - p2: both terms move, both are bf16-rounded in the oracle
    else:
        base = conf_o.abs()
    worst = (grow + 2.0 ** -7) * base + 1e-4
    if sens is None:
        return worst
    tight = 0.5 * grow * sens + 2.0 ** -7 * (conf_o.abs() if alg != "topk_margin" else 2.0 * pmax) + 1e-4
    return torch.minimum(worst, tight)


def hash_uniform_host(seed: int, rowkey: torch.Tensor, v) -> torch.Tensor:
    """Host restatement of coda_hash_uniform (csrc/coda_common.h): uniform in (0, 1) from (seed, row key, index).
    ``rowkey`` and ``v`` broadcast (int64 tensors or ints)."""
    M = 0xFFFFFFFF
    lo, hi = seed & M, (seed >> 32) & M
    v = torch.as_tensor(v, dtype=torch.int64)
    h = ((v * 0x9E3779B1) & M) ^ ((torch.as_tensor(rowkey, dtype=torch.int64) * 0x85EBCA77) & M) ^ lo
    h = h ^ (h >> 16); h = (h * 0x85EBCA6B) & M; h = h ^ (h >> 13); h = (h * 0xC2B2AE35) & M; h = h ^ (h >> 16)
    h = (h + hi) & M
    h = h ^ (h >> 15); h = (h * 0x2C1B3C6D) & M; h = h ^ (h >> 12); h = (h * 0x297A2D39) & M; h = h ^ (h >> 15)
    return ((h >> 9).to(torch.float32) + 0.5) * (1.0 / 8388608.0)


def subsample(x_init, hist, trace, rows, steps):
    """Cuts rows ``rows`` and denoise steps ``steps`` out of a full-size run (history + debug trace) and re-indexes the
    trace's flat positions to the sub-batch, for teacher_forced(states=..., step_indices=...)."""
    L = x_init.shape[1]
    states = [(x_init if i == 0 else hist[i - 1])[rows] for i in steps]
    sub_hist = [hist[i][rows] for i in steps]
    sub_trace = []
    for i in steps:
        t = trace[i]
        n = int(t["count"].item())
        dst = t["dst"][:n].long().cpu()
        conf, x0 = t["conf"].cpu(), t["x0"].cpu()
        row_of = dst // L
        keep = torch.isin(row_of, torch.tensor(rows)).nonzero().squeeze(1)
        new_row = torch.tensor([rows.index(int(r)) for r in row_of[keep]], dtype=torch.long)
        new_dst = (new_row * L + dst[keep] % L).to(torch.int32)
        c2 = torch.zeros(len(rows) * L)
        z2 = torch.zeros(len(rows) * L, dtype=torch.int64)
        c2[new_dst.long()] = conf[dst[keep]]
        z2[new_dst.long()] = x0[dst[keep]]
        sub_trace.append(dict(count=torch.tensor([len(keep)]), dst=new_dst, conf=c2, x0=z2, orig_dst=dst[keep]))
    return states, sub_hist, sub_trace


def describe(st) -> str:
    """One line for test output: the error is printed next to the tolerance it was held against."""
    n = max(st["candidates"], 1)
    if st["conf_err"]:
        err, tol = torch.cat(st["conf_err"]), torch.cat(st["conf_tol"])
        conf = (f"|conf err| mean {float(err.mean()):.4g} / max {st['max_conf_err']:.4g} against tolerances mean {float(tol.mean()):.3g} / "
                f"at the worst point {st['tol_at_max']:.3g} (worst err/tol {st['max_conf_err_over_tol']:.2f})")
    else:
        conf = "no confidences compared"
    return (f"{st['exact_steps']}/{st['steps']} steps identical to the bf16 oracle; {st['candidates']} candidate rows, "
            f"{st['band_tok']} tokens ({100.0 * st['band_tok'] / n:.1f} %) inside the declared band at FWD_ERR_ULPS {st['fwd_err_ulps']:g}"
            + (f" ({st['band_tok_wide']} = {100.0 * st['band_tok_wide'] / n:.1f} % of them beyond the {FWD_ERR_ULPS_SMALL:g}-ulp band)"
               if st["fwd_err_ulps"] > FWD_ERR_ULPS_SMALL else "")
            + (f", {st['band_nucleus']} by nucleus membership" if st.get("band_nucleus") else "")
            + f"; {conf}; {len(st['violations'])} violations")


def assert_parity(st, max_excused: float, min_candidates: int = 1, max_wide: float = 0.03):
    """No violations, and the share of candidates whose token differs INSIDE the declared band stays below a stated bound
    (a band that excuses too much hides bugs: ADVICE round 2).  ``max_wide`` bounds the share that only the full-size band
    (FWD_ERR_ULPS_FULL) excuses and the small-model band would not: per logit about 1 in 12 deviations exceeds 1.5 ulps at 28
    layers, and it must also meet a top-2 distance of that size, so a few per cent is the measured rate."""
    assert st["candidates"] >= min_candidates, st["candidates"]
    assert not st["violations"], st["violations"][:5]
    n = max(st["candidates"], 1)
    assert st["band_tok"] / n <= max_excused, f"{st['band_tok']} of {n} tokens excused by the band (> {max_excused})"
    assert st["band_tok_wide"] / n <= max_wide, f"{st['band_tok_wide']} of {n} tokens excused only by the wide band (> {max_wide})"


def teacher_forced(sd, d, ids, hip_hist, hip_trace, noise_log=None, states=None, step_indices=None,
                   fwd_err_ulps: float = FWD_ERR_ULPS_SMALL, **kw):
    """``states`` / ``step_indices`` (optional): compare only the given steps - hip_hist[n] / hip_trace[n] are then the
    HIP path's state after / trace of step step_indices[n] and states[n] its state before it (full-size sample)."""
    G = kw["max_new_tokens"]
    alg, alg_temp, temperature = kw["alg"], kw.get("alg_temp") or 0.0, kw.get("temperature") or 0.0
    x_init = F.pad(ids, (0, G), value=d.mask_id)
    if states is None:
        states = [x_init] + list(hip_hist[:-1])
    noise = ref_cpu.ReplayNoise(noise_log) if noise_log is not None else None
    tr = ref_cpu.diffusion_generate(sd, d, ids, force_states=states, trace=True, noise=noise, step_indices=step_indices, **kw)
    sel_noise = {s: t for k_, s, t in (noise_log or []) if k_ == "select"}
    E = float(fwd_err_ulps)
    st = dict(steps=len(hip_hist), exact_steps=0, candidates=0, band_tok=0, conf_checked=0, max_conf_err=0.0,
              max_conf_err_over_tol=0.0, tol_at_max=0.0, max_band=0.0, violations=[], fwd_err_ulps=E,
              band_tok_wide=0, conf_err=[], conf_tol=[])
    B, L = x_init.shape
    if alg == "confidence_threshold":
        # the serving sampler's bookkeeping (serving/dream/model/generation_utils.py:416-465), replayed on the HIP
        # path's own confidences: quota per step, tokens held back, and the data-dependent step count
        thr_steps = kw["steps"]
        thr_quota, thr_left = G // thr_steps, 0
        threshold = kw.get("threshold", 0.9)
    for i, new in enumerate(hip_hist):
        prev, cand = states[i], tr.cand[i]
        if torch.equal(tr.history[i], new):
            st["exact_steps"] += 1
        n = int(cand.sum())
        st["candidates"] += n
        ht = hip_trace[i]
        if int(ht["count"].item()) != n:
            st["violations"].append((i, f"candidate count {int(ht['count'].item())} != {n}"))
            continue
        flat = cand.reshape(-1).nonzero().squeeze(1)
        if not torch.equal(ht["dst"][:n].long().cpu(), flat):
            st["violations"].append((i, "candidate order differs"))
            continue
        conf_h = ht["conf"].cpu()[flat]
        x0_h = ht["x0"].cpu()[flat]
        x0_o = tr.x0[i]
        if alg == "origin":
            picked = x0_o != d.mask_id                      # the oracle only samples the transferred rows
            gap = tr.gap[i]
            tok_bad = picked & (x0_h != x0_o)
            band = torch.full_like(gap, BAND_TOK) if temperature <= 0 else torch.full_like(gap, -math.expm1(-4 * 0.125 / temperature))
            st["band_tok"] += int((tok_bad & (gap <= band)).sum())
            if (tok_bad & (gap > band)).any():
                st["violations"].append((i, f"token differs outside band, gaps {gap[tok_bad].tolist()}"))
            expect = prev.clone().reshape(-1)
            expect[flat[picked]] = x0_h[picked]
            if not torch.equal(expect.reshape(B, L), new):
                st["violations"].append((i, "origin transfer set differs"))
            continue
        # ---- tokens
        gap = tr.gap[i]
        if temperature <= 0 and tr.gap_abs[i] is not None:
            gap = tr.gap_abs[i]                       # greedy: distance of the two largest logits, in logit units
        tok_bad = x0_h != x0_o
        band = token_band(tr.absmax[i], temperature, E)
        excused = gap <= band
        narrow = gap <= token_band(tr.absmax[i], temperature, min(E, FWD_ERR_ULPS_SMALL))
        top_p = kw.get("top_p")
        if top_p is not None and top_p < 1 and tr.nucleus[i] is not None and n:
            # top-p: the race runs over the nucleus only, and membership is a threshold on cumulative mass.  A token is
            # "on the boundary" when the mass in front of it (oracle's order), scaled by what the logit band does to
            # probabilities (e^+-u), crosses top_p; the oracle also compares a bf16-rounded cumsum with a bf16-rounded
            # top_p (one bf16 step near 0.9 = 2^-8).  The oracle's token may then be outside the HIP path's nucleus, or
            # the HIP path's token outside the oracle's: either explains a different winner whatever the race gap.
            u = logit_band(tr.absmax[i], temperature, E)
            ar = torch.arange(n)
            c_o, c_h = tr.nucleus[i][ar, x0_o], tr.nucleus[i][ar, x0_h.clamp(0, tr.nucleus[i].shape[1] - 1)]
            boundary = (c_o * torch.exp(u) >= top_p - 2.0 ** -8) | ((c_h > top_p - 2.0 ** -8) & (c_h * torch.exp(-u) <= top_p + 2.0 ** -8))
            st["band_nucleus"] = st.get("band_nucleus", 0) + int((tok_bad & ~excused & boundary).sum())
            excused = excused | boundary
            u_n = logit_band(tr.absmax[i], temperature, min(E, FWD_ERR_ULPS_SMALL))
            narrow = narrow | (c_o * torch.exp(u_n) >= top_p - 2.0 ** -8) | ((c_h > top_p - 2.0 ** -8) & (c_h * torch.exp(-u_n) <= top_p + 2.0 ** -8))
        st["band_tok"] += int((tok_bad & excused).sum())
        st["band_tok_wide"] += int((tok_bad & excused & ~narrow).sum())      # excused only by the full-size band
        if (tok_bad & ~excused).any():
            st["violations"].append((i, f"token differs outside band, gaps {gap[tok_bad & ~excused].tolist()}"))
        # ---- confidences (only where the token agrees: with sampling the confidence is p[x0])
        ok = ~tok_bad
        conf_o = tr.conf[i].float()
        filtered = (top_p is not None and top_p < 1) or kw.get("top_k") is not None
        tol = conf_tolerance(conf_o, tr.absmax[i], tr.pmax[i], "maskgit_plus" if alg == "confidence_threshold" else alg, temperature,
                             E, None if filtered else tr.sens[i])
        err = (conf_h - conf_o).abs()
        if ok.any():
            st["conf_err"].append(err[ok]); st["conf_tol"].append(tol[ok])
            st["conf_checked"] += int(ok.sum())
            st["max_conf_err"] = max(st["max_conf_err"], float(err[ok].max()))
            ratio = (err / tol).masked_fill(~ok, -1.0)
            if float(ratio.max()) > st["max_conf_err_over_tol"]:
                st["max_conf_err_over_tol"] = float(ratio.max())
                st["tol_at_max"] = float(tol[int(ratio.argmax())])
            st["max_band"] = max(st["max_band"], float(band.max()))
            if (err[ok] > tol[ok]).any():
                j = int((err / tol).masked_fill(~ok, 0).argmax())
                st["violations"].append((i, f"confidence outside tolerance: hip {conf_h[j]:.6g} oracle {conf_o[j]:.6g} tol {tol[j]:.3g}"))
        # ---- selection, exact, from the HIP path's own confidences
        if alg == "confidence_threshold":
            cur = thr_quota + thr_left
            pos = flat.tolist()
            order = sorted(range(n), key=lambda c: (-float(conf_h[c]), pos[c]))[:cur]
            keep = [c for r, c in enumerate(order) if r == 0 or not (float(conf_h[c]) < threshold)]
            thr_left = len(order) - len(keep)
            if thr_left and i >= thr_steps - 1:
                thr_quota, thr_steps = 0, thr_steps + 1
            expect = prev.clone().reshape(-1)
            for c in keep:
                expect[pos[c]] = x0_h[c]
            if not torch.equal(expect.reshape(B, L), new):
                st["violations"].append((i, f"threshold selection differs from the reference rule on the HIP confidences (k={cur})"))
            if i == len(hip_hist) - 1 and thr_steps != len(hip_hist):
                st["violations"].append((i, f"step count {len(hip_hist)} != {thr_steps} implied by the reference bookkeeping"))
            continue
        k = tr.k[i]
        full = torch.full((B * L,), -torch.inf)
        full[flat] = conf_h
        full = full.reshape(B, L)
        key = full
        if alg_temp > 0:
            key = torch.softmax(full / alg_temp, -1) / sel_noise[step_indices[i] if step_indices is not None else i].float()
            key = key.masked_fill(~cand, -torch.inf)
        expect = prev.clone()
        x0_full = torch.full((B * L,), -1, dtype=torch.long)
        x0_full[flat] = x0_h
        x0_full = x0_full.reshape(B, L)
        near = False
        for b in range(B):
            pos = cand[b].nonzero().squeeze(1).tolist()
            order = sorted(pos, key=lambda p: (-float(key[b, p]), p))[:max(k, 0)]
            expect[b, order] = x0_full[b, order]
            if 0 < k < len(pos):
                srt = sorted((float(key[b, p]) for p in pos), reverse=True)
                if abs(srt[k - 1] - srt[k]) <= 1e-5 * max(abs(srt[k - 1]), 1e-30):
                    near = True
        if not torch.equal(expect, new) and not near:
            st["violations"].append((i, f"selection differs from the reference rule on the HIP confidences (k={k})"))
    return st, tr
