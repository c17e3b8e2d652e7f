"""numpy ORACLE for the PBCH/MIB decode (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Restates what lib/mib_impl.cc:148-170 asks of srsLTE 18.06.1 (`srslte_ue_mib_decode`,
`srslte_pbch_mib_unpack`) from 3GPP TS 36.211 6.6 (PBCH), 6.10.1 (CRS), 7.2 (Gold sequence) and
TS 36.212 5.1.1 (CRC16), 5.1.3.1 (tail-biting convolutional code), 5.1.4.2 (rate matching), 5.3.1.
Decoding is single-frame only because the block calls srslte_pbch_decode_reset before every
attempt (lib/mib_impl.cc:161).  Pinned by the MIB fields the reference's tests assert on the four
fixtures (python/qa_downlink_trigger_c.py:93-98 etc.): nof_prb, nof_tx_ports, phich_len,
nof_phich_resources.  Only tests/ may import this module.
"""
import numpy as np

FFT = 128
SF_LEN = 1920
NOF_RE = 72  # 6 PRB

BW_TABLE = [6, 15, 25, 50, 75, 100]
PHICH_R = ["1/6", "1/2", "1", "2"]
CRC_MASK = {1: 0x0000, 2: 0xFFFF, 4: 0x5555}


def gold(c_init, n):
    """36.211 7.2 pseudo-random sequence c(n)."""
    nc = 1600
    x1 = np.zeros(nc + n + 31, np.uint8)
    x2 = np.zeros(nc + n + 31, np.uint8)
    x1[0] = 1
    for i in range(31):
        x2[i] = (c_init >> i) & 1
    for i in range(nc + n):
        x1[i + 31] = x1[i + 3] ^ x1[i]
        x2[i + 31] = x2[i + 3] ^ x2[i + 2] ^ x2[i + 1] ^ x2[i]
    return x1[nc:nc + n] ^ x2[nc:nc + n]


def symbol_start(l, cp_ext):
    """Offset of the FFT window of symbol l inside a 960-sample slot at 1.92 Msps."""
    if cp_ext:
        return l * (FFT + 32) + 32
    return 10 if l == 0 else 138 + (l - 1) * 137 + 9


def ofdm_symbol(slot, l, cp_ext):
    """72 centre subcarriers of OFDM symbol l of the slot (k = 0..71 <-> bins -36..-1, +1..+36)."""
    s = symbol_start(l, cp_ext)
    Y = np.fft.fft(slot[s:s + FFT].astype(np.complex128))
    return np.concatenate([Y[FFT - 36:], Y[1:37]])


def crs_values(cell_id, ns, l, cp_ext):
    """r_{l,ns}(m') for the 12 CRS REs of one port in a 6-PRB symbol (36.211 6.10.1.1)."""
    n_cp = 0 if cp_ext else 1
    c_init = (1 << 10) * (7 * (ns + 1) + l + 1) * (2 * cell_id + 1) + 2 * cell_id + n_cp
    c = gold(c_init, 2 * 220).astype(np.float64)
    m = np.arange(12) + 110 - 6
    return ((1 - 2 * c[2 * m]) + 1j * (1 - 2 * c[2 * m + 1])) / np.sqrt(2)


def crs_positions(cell_id, port, ns, l, cp_ext):
    """Subcarrier indices k = 6m + (v + v_shift) mod 6 of the CRS of `port` in symbol l, or None."""
    nsymb = 6 if cp_ext else 7
    if port in (0, 1):
        if l == 0:
            v = 0 if port == 0 else 3
        elif l == nsymb - 3:
            v = 3 if port == 0 else 0
        else:
            return None
    else:
        if l != 1:
            return None
        v = 3 * (ns % 2) if port == 2 else 3 + 3 * (ns % 2)
    return 6 * np.arange(12) + (v + cell_id % 6) % 6


def channel_estimate(sf, cell_id, port, cp_ext):
    """LS estimates on the CRS of slot 1, merged over the slot's CRS symbols and interpolated
    linearly over frequency (static channel over the four PBCH symbols)."""
    slot1 = sf[960:1920]
    nsymb = 6 if cp_ext else 7
    ks, hs = [], []
    for l in range(nsymb):
        pos = crs_positions(cell_id, port, 1, l, cp_ext)
        if pos is None:
            continue
        y = ofdm_symbol(slot1, l, cp_ext)
        ks.append(pos)
        hs.append(y[pos] * np.conj(crs_values(cell_id, 1, l, cp_ext)))
    k = np.concatenate(ks)
    h = np.concatenate(hs)
    order = np.argsort(k, kind="stable")
    k, h = k[order], h[order]
    grid = np.arange(NOF_RE)
    return np.interp(grid, k, h.real) + 1j * np.interp(grid, k, h.imag)


def pbch_re_indices(cell_id, l, cp_ext):
    """PBCH REs of symbol l of slot 1: every RE except those reserved for CRS of ports 0..3."""
    k = np.arange(NOF_RE)
    crs_syms = (0, 1, 3) if cp_ext else (0, 1)
    if l in crs_syms:
        return k[(k % 3) != (cell_id % 3)]
    return k


SUBBLOCK_PERM = [1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31,
                 0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30]


def rate_match_pattern(D=40):
    """36.212 5.1.4.2: index into d (stream*D + pos) of every non-null bit of the circular buffer."""
    C = 32
    R = (D + C - 1) // C
    nd = R * C - D
    out = []
    for stream in range(3):
        for col in SUBBLOCK_PERM:
            for row in range(R):
                idx = row * C + col - nd
                if idx >= 0:
                    out.append(stream * D + idx)
    return np.array(out)


_VIT = None


def _viterbi_tables():
    """Trellis of the (133,171,165) code: for every next state its two predecessors (even one first), the input bit
    that leads there and the sign (+1: coded bit 0) of the three coded bits on each of the two branches."""
    global _VIT
    if _VIT is None:
        polys = [0o133, 0o171, 0o165]
        ns = np.arange(64)
        b = ns >> 5                      # input bit: the shift register's newest entry
        s0 = (ns & 31) << 1              # predecessors s0, s0 + 1: next = (s >> 1) | (b << 5)
        sign = np.zeros((2, 64, 3))
        for j in range(2):
            for n in range(64):
                reg = (int(b[n]) << 6) | (int(s0[n]) + j)
                for i, p in enumerate(polys):
                    sign[j, n, i] = 1.0 - 2.0 * (bin(reg & p).count("1") & 1)
        _VIT = (s0, b.astype(np.int8), sign)
    return _VIT


def viterbi_tailbiting(llr, nbits=40):
    """Rate-1/3 K=7 (133,171,165) tail-biting Viterbi on soft bits (positive = 0); llr is
    [nbits, 3].  Three passes over the circular trellis, decisions taken from the middle pass.
    All 64 states of a step at once; a state reached with equal metrics from both predecessors keeps the even one
    (the first a state-by-state loop with a strict comparison would meet)."""
    s0, bit, sign = _viterbi_tables()
    passes = 3
    T = nbits * passes
    metric = np.zeros(64)
    prev = np.zeros((T, 64), np.int64)
    llr = np.asarray(llr, dtype=np.float64)
    for t in range(T):
        l = llr[t % nbits]
        bm = (sign[:, :, 0] * l[0] + sign[:, :, 1] * l[1]) + sign[:, :, 2] * l[2]  # branch metrics, summed in order
        m0 = metric[s0] + bm[0]
        m1 = metric[s0 + 1] + bm[1]
        odd = m1 > m0
        new = np.where(odd, m1, m0)
        prev[t] = s0 + odd
        metric = new - new.max()
    s = int(np.argmax(metric))
    bits = np.zeros(T, np.uint8)
    for t in range(T - 1, -1, -1):
        bits[t] = bit[s]
        s = prev[t, s]
    return bits[nbits:2 * nbits]


def crc16(bits):
    reg = 0
    for b in bits:
        fb = ((reg >> 15) & 1) ^ int(b)
        reg = (reg << 1) & 0xFFFF
        if fb:
            reg ^= 0x1021
    return reg


def equalise(ys, hs, nof_ports):
    """PBCH symbols -> 240 (or 216) estimates of the transmitted layer symbols.
    ys: received PBCH REs in order; hs[p]: channel of port p on those REs (36.211 6.3.3/6.3.4)."""
    if nof_ports == 1:
        h = hs[0]
        return ys * np.conj(h) / np.maximum(np.abs(h) ** 2, 1e-12)
    x = np.zeros(ys.size, complex)
    for i in range(0, ys.size - 1, 2):
        if nof_ports == 2:
            pa, pb = 0, 1
        else:  # 4 ports: pairs alternate between antenna pairs (0,2) and (1,3) (SFBC + FSTD)
            pa, pb = (0, 2) if (i // 2) % 2 == 0 else (1, 3)
        h0 = 0.5 * (hs[pa][i] + hs[pa][i + 1])
        h1 = 0.5 * (hs[pb][i] + hs[pb][i + 1])
        r0, r1 = ys[i], ys[i + 1]
        den = max(abs(h0) ** 2 + abs(h1) ** 2, 1e-12)
        # y0 = (h0 x0 - h1 x1*)/sqrt2 ; y1 = (h0 x1 + h1 x0*)/sqrt2
        x[i] = (np.conj(h0) * r0 + h1 * np.conj(r1)) / den * np.sqrt(2)
        x[i + 1] = (-h1 * np.conj(r0) + np.conj(h0) * r1) / den * np.sqrt(2)
    return x


def decode_mib(subframe, cell_id, cp_ext=False, ports_to_try=(1, 2, 4)):
    """subframe: 1920 cf32 samples starting at the slot-0 boundary of a subframe-0 candidate.
    Returns dict(nof_prb, nof_tx_ports, phich_len, nof_phich_resources, sfn_offset, bits, frame_idx)
    or None."""
    sf = np.asarray(subframe[:SF_LEN], dtype=np.complex64)
    slot1 = sf[960:1920]
    ys = []
    idxs = []
    for l in range(4):
        y = ofdm_symbol(slot1, l, cp_ext)
        k = pbch_re_indices(cell_id, l, cp_ext)
        ys.append(y[k])
        idxs.append(k)
    ys = np.concatenate(ys)
    nbits = 2 * ys.size
    scr = gold(cell_id, 1920)
    pattern = rate_match_pattern(40)
    H = {}
    for nports in ports_to_try:
        for p in range(nports):
            if p not in H:
                h = channel_estimate(sf, cell_id, p, cp_ext)
                H[p] = np.concatenate([h[k] for k in idxs])
        x = equalise(ys, [H[p] for p in range(nports)], nports)
        soft = np.empty(nbits)
        soft[0::2] = x.real
        soft[1::2] = x.imag
        for frame_idx in range(4):
            seg = scr[nbits * frame_idx: nbits * frame_idx + nbits].astype(np.float64)
            d = soft * (1 - 2 * seg)
            acc = np.zeros(120)
            np.add.at(acc, pattern[(nbits * frame_idx + np.arange(nbits)) % 120], d)
            bits = viterbi_tailbiting(acc.reshape(3, 40).T)
            crc = 0
            for b in bits[24:]:
                crc = (crc << 1) | int(b)
            if (crc16(bits[:24]) ^ CRC_MASK[nports]) == crc and np.any(bits[:24]):
                bw = int(bits[0]) << 2 | int(bits[1]) << 1 | int(bits[2])
                if bw >= len(BW_TABLE):
                    continue
                sfn = 0
                for b in bits[6:14]:
                    sfn = (sfn << 1) | int(b)
                return dict(nof_prb=BW_TABLE[bw], nof_tx_ports=nports,
                            phich_len="Extended" if bits[3] else "Normal",
                            nof_phich_resources=PHICH_R[int(bits[4]) << 1 | int(bits[5])],
                            sfn_offset=sfn << 2,  # srslte_pbch_mib_unpack overwrites d_sfn_offset, lib/mib_impl.cc:168-170
                            frame_idx=frame_idx, bits="".join(str(int(b)) for b in bits[:24]))
    return None
