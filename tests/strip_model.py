"""numpy model of the strip-wise search the scan kernel runs (csrc/lt_scan.h).

srslte_pss_find_pss downstream of the EMA (arg-max, main-lobe walk, side-lobe maximum, PSR; restated in
oracle/lte_oracle.c:272-317) needs the whole 9726-lag EMA vector of a window.  The scan kernel holds that
vector in 12 lag strips spread over four workgroups, so every strip reduces its own lags to a small partial
result and a resolver combines the 12 partials of a window.  This file restates both halves in numpy, lag
for lag as the wave code does it, so that the CPU suite can check the decomposition against the direct
algorithm on random and on adversarial vectors (tests/test_strip_model.py).

Geometry (the same constants as csrc/lt_scan.h):
  strip s owns lags [811 s, 811 s + 811) and computes lags c0 .. c0 + 895, c0 = 811 s - 41
  lane l, register r holds lag c0 + l + 64 r (r < 14); lags < 0 or >= 9726 hold 0
"""
import numpy as np

NLAGS = 9726
CONV_LEN = 9728
PITCH = 811
STRIPS = 12
HALO = 41
REGS = 14
FLT_MAX = np.float32(np.finfo(np.float32).max)

F_WIDE = 1  # lobe not closed inside the 64-lag neighbourhood of the strip's peak


def find_reference(avg):
    """oracle/lte_oracle.c:295-316 on a 9729-float EMA vector (entries >= 9726 are 0)."""
    L = NLAGS
    peak = 0
    m = -FLT_MAX
    for i in range(L):  # srslte_vec_max_fi: first strict maximum
        if avg[i] > m:
            m = avg[i]
            peak = i
    pl_ub = peak + 1
    while avg[pl_ub + 1] <= avg[pl_ub] and pl_ub < CONV_LEN - 1:
        pl_ub += 1
    if peak > 2:
        pl_lb = peak - 1
        while avg[pl_lb - 1] <= avg[pl_lb] and pl_lb > 1:
            pl_lb -= 1
    else:
        pl_lb = 0
    dist_r = max(CONV_LEN - 2 - pl_ub, 0)

    def vmax(x):
        mm, p = -FLT_MAX, 0
        for i, v in enumerate(x):
            if v > mm:
                mm, p = v, i
        return p
    sl_right = pl_ub + vmax(avg[pl_ub:pl_ub + dist_r])
    sl_left = vmax(avg[:pl_lb])
    side = avg[sl_left] if avg[sl_left] > avg[sl_right] else avg[sl_right]
    with np.errstate(divide="ignore", invalid="ignore"):
        psr = np.float32(avg[peak]) / np.float32(side)
    return peak, np.float32(avg[peak]), psr


def strip_regs(avg, s):
    """The strip's register image: e[r][l] = EMA of lag c0 + l + 64 r (0 outside [0, 9726))."""
    c0 = PITCH * s - HALO
    lag = c0 + np.arange(64)[None, :] + 64 * np.arange(REGS)[:, None]
    ok = (lag >= 0) & (lag < NLAGS)
    e = np.where(ok, avg[np.clip(lag, 0, CONV_LEN)], np.float32(0)).astype(np.float32)
    return e, lag


def strip_partial(avg, s):
    """What one root wave of the scan kernel reduces its strip to (StripPartial in csrc/lt_scan.h)."""
    e, lag = strip_regs(avg, s)
    c0 = PITCH * s - HALO
    off = np.arange(64)[None, :] + 64 * np.arange(REGS)[:, None]
    own = (off >= HALO) & (off < HALO + PITCH) & (lag < NLAGS)
    # 1. maximum over the owned lags, first in lag order (smallest register, then smallest lane)
    masked = np.where(own, e, -FLT_MAX)
    m1 = masked.max()
    r_p = l_p = -1
    for r in range(REGS):
        hit = np.nonzero(masked[r] == m1)[0]
        if hit.size:
            r_p, l_p = r, int(hit[0])
            break
    p1 = c0 + l_p + 64 * r_p
    # 2. the 64 lags p1 - 32 .. p1 + 31 as a rotated row: lane L holds offset (L - l_p) wrapped to [-32, 32)
    r_b = r_p - 1 if l_p < 32 else r_p + 1
    L = np.arange(64)
    use_b = (L >= l_p + 32) if l_p < 32 else (L < l_p - 32)
    row = np.where(use_b, e[min(max(r_b, 0), REGS - 1)], e[r_p])
    nxt = np.roll(row, -1)  # value of lane L + 1
    prv = np.roll(row, 1)   # value of lane L - 1
    d_r = nxt <= row        # a[u + 1] <= a[u]
    d_l = prv <= row        # a[l - 1] <= a[l]
    # walk with the window-global bounds of oracle/lte_oracle.c:300-308 folded in: the right walk also stops
    # at lag 9727, the left walk at lag 1; a peak at lag <= 2 has no left walk at all (pl_lb = 0)
    off_row = ((L - l_p + 32) % 64) - 32          # wrapped offset of lane L
    lag_row = p1 + off_row
    stop_r = ~d_r | (lag_row >= CONV_LEN - 1)
    stop_l = ~d_l | (lag_row <= 1)
    flags = 0
    j_r = None
    for j in range(1, 31):
        if stop_r[(l_p + j) % 64]:
            j_r = j
            break
    j_l = None
    if p1 > 2:
        for j in range(1, 32):
            if stop_l[(l_p - j) % 64]:
                j_l = j
                break
    else:
        j_l = p1  # lb = 0
    if j_r is None or j_l is None:
        return dict(m1=m1, p1=p1, s=np.float32(0), lb=0, ub=0, a0=np.float32(e[0][HALO] if s == 0 else 0), flags=F_WIDE)
    lb, ub = p1 - j_l, p1 + j_r
    # 3. maximum over the owned lags outside [lb, ub)
    outside = own & ~((lag >= lb) & (lag < ub))
    sv = np.where(outside, e, -FLT_MAX).max()
    a0 = np.float32(e[0][HALO]) if s == 0 else np.float32(0)
    return dict(m1=np.float32(m1), p1=int(p1), s=np.float32(sv), lb=int(lb), ub=int(ub), a0=a0, flags=flags)


def combine(parts):
    """The resolver's per-window function (resolve_window in csrc/lt_scan.h): 12 partials -> peak, PSR.
    Returns None when the window has to go to the exact path (tracker kernel)."""
    best = 0
    for s in range(1, STRIPS):  # strips ascend in lag: strict > keeps the first maximum
        if parts[s]["m1"] > parts[best]["m1"]:
            best = s
    pk = parts[best]
    if pk["flags"] & F_WIDE:
        return None
    peak, peak_value = pk["p1"], pk["m1"]
    if not peak_value > -FLT_MAX:
        peak = 0
    lb_l, ub_l = pk["lb"], pk["ub"]
    out = pk["s"]
    for s in range(STRIPS):
        if s == best:
            continue
        q = parts[s]
        inside = lb_l <= q["p1"] < ub_l
        if inside and (q["flags"] & F_WIDE):
            return None
        v = q["s"] if inside else q["m1"]
        if v > out:
            out = v
    a0 = parts[0]["a0"]
    # empty ranges stand for their first element (srslte_vec_max_fi returns index 0): the left one is empty
    # exactly when the peak sits at lag <= 2, the right one when the lobe reaches lag 9726 (avg[9726..] = 0)
    sl = a0 if peak <= 2 else out
    sr = np.float32(0) if ub_l >= NLAGS else out
    side = sl if sl > sr else sr
    with np.errstate(divide="ignore", invalid="ignore"):
        psr = np.float32(peak_value) / np.float32(side)
    return peak, np.float32(peak_value), psr


def find_strips(avg):
    return combine([strip_partial(avg, s) for s in range(STRIPS)])


# ---------------------------------------------------------------- the deferred form (scan rounds since round 5)
# A root wave stops after the strip's maximum: it leaves a head (m1, p1, far = the maximum over its owned lags outside
# the 32 lags around p1, a0) and the row of those 32 lags; the resolver walks the lobe of the one strip that holds the
# window's peak (strip_head_row / resolve_window_deferred in csrc/lt_scan.h).  A lobe that does not close within 14 lags
# to the right or 15 to the left sends the window to the exact path.
ROW = 32
HALF = ROW // 2


def strip_head_row(avg, s):
    e, lag = strip_regs(avg, s)
    c0 = PITCH * s - HALO
    off = np.arange(64)[None, :] + 64 * np.arange(REGS)[:, None]
    own = (off >= HALO) & (off < HALO + PITCH) & (lag < NLAGS)
    masked = np.where(own, e, -FLT_MAX)
    m1 = masked.max()
    r_p = l_p = -1
    for r in range(REGS):
        hit = np.nonzero(masked[r] == m1)[0]
        if hit.size:
            r_p, l_p = r, int(hit[0])
            break
    p1 = c0 + l_p + 64 * r_p
    # per lane: maximum over its owned lags, and its second largest (with multiplicity: a tie gives m2 == m)
    srt = np.sort(masked, axis=0)
    m, m2 = srt[-1], srt[-2]
    r_b = (r_p - 1 if r_p > 0 else 1) if l_p < 32 else (r_p + 1 if r_p < REGS - 1 else REGS - 2)
    L = np.arange(64)
    d = L - l_p
    offs = ((d + 32) % 64) - 32
    use_b = d != offs
    row = np.where(use_b, e[r_b], e[r_p])
    row_own = np.where(use_b, masked[r_b], masked[r_p])  # -FLT_MAX where the lag is not the strip's own
    in_row = (offs >= -HALF) & (offs < HALF)
    far_lane = np.where(in_row & (row_own == m), m2, m)
    out_row = np.zeros(ROW, np.float32)
    out_row[offs[in_row] + HALF] = row[in_row]
    a0 = np.float32(e[0][HALO]) if s == 0 else np.float32(0)
    return dict(m1=np.float32(m1), p1=int(p1), far=np.float32(far_lane.max()), a0=a0, row=out_row, s=s)


def _owned(s, lag):
    return PITCH * s <= lag < PITCH * (s + 1) and lag < NLAGS


def _outside(h, lb, ub):
    v = h["far"]
    for i in range(ROW):
        lag = h["p1"] - HALF + i
        if _owned(h["s"], lag) and (lag < lb or lag >= ub) and h["row"][i] > v:
            v = h["row"][i]
    return v


def combine_deferred(heads):
    best = 0
    for s in range(1, STRIPS):
        if heads[s]["m1"] > heads[best]["m1"]:
            best = s
    hb = heads[best]
    p1, r = hb["p1"], hb["row"]
    peak_value = hb["m1"]
    peak = p1 if peak_value > -FLT_MAX else 0
    j_r = next((j for j in range(1, HALF - 1) if (not r[HALF + j + 1] <= r[HALF + j]) or p1 + j >= CONV_LEN - 1), None)
    if p1 <= 2:
        j_l = p1
    else:
        j_l = next((j for j in range(1, HALF) if (not r[HALF - j - 1] <= r[HALF - j]) or p1 - j <= 1), None)
    if j_r is None or j_l is None:
        return None
    lb, ub = p1 - j_l, p1 + j_r
    out = _outside(hb, lb, ub)
    for s in range(STRIPS):
        if s == best:
            continue
        q = heads[s]
        v = _outside(q, lb, ub) if lb <= q["p1"] < ub else q["m1"]
        if v > out:
            out = v
    sl = heads[0]["a0"] if peak <= 2 else out
    sr = np.float32(0) if ub >= NLAGS else out
    side = sl if sl > sr else sr
    with np.errstate(divide="ignore", invalid="ignore"):
        psr = np.float32(peak_value) / np.float32(side)
    return peak, np.float32(peak_value), psr


def find_strips_deferred(avg):
    return combine_deferred([strip_head_row(avg, s) for s in range(STRIPS)])
