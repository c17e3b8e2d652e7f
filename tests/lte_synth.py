"""Synthetic LTE FDD downlink frames at 1.92 Msps (6 PRB view), for tests only.

Independent of the oracle and of the product: written from 3GPP TS 36.211 (6.3 layer mapping /
precoding, 6.6 PBCH, 6.10.1 CRS, 6.11 PSS/SSS, 6.12 OFDM) and TS 36.212 (5.1.1, 5.1.3.1, 5.1.4.2).
Generates PSS, SSS, CRS and PBCH for any of the 504 cell ids, normal or extended CP, 1/2/4 ports.
"""
import numpy as np

FFT = 128
NRE = 72


def gold(c_init, n):
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


def pss_seq(n_id_2):
    u = [25, 29, 34][n_id_2]
    n = np.arange(62)
    return np.where(n < 31, np.exp(-1j * np.pi * u * n * (n + 1) / 63),
                    np.exp(-1j * np.pi * u * (n + 1) * (n + 2) / 63))


def _mseq(taps):
    x = [0, 0, 0, 0, 1]
    for i in range(26):
        x.append(sum(x[i + t] for t in taps) % 2)
    return 1 - 2 * np.array(x)


def sss_seq(n_id_1, n_id_2, subframe):
    qp = n_id_1 // 30
    q = (n_id_1 + qp * (qp + 1) // 2) // 30
    mp = n_id_1 + q * (q + 1) // 2
    m0 = mp % 31
    m1 = (m0 + mp // 31 + 1) % 31
    s_t, c_t, z_t = _mseq([2, 0]), _mseq([3, 0]), _mseq([4, 2, 1, 0])
    n = np.arange(31)
    s0, s1 = s_t[(n + m0) % 31], s_t[(n + m1) % 31]
    c0, c1 = c_t[(n + n_id_2) % 31], c_t[(n + n_id_2 + 3) % 31]
    z1m0, z1m1 = z_t[(n + (m0 % 8)) % 31], z_t[(n + (m1 % 8)) % 31]
    d = np.zeros(62)
    if subframe == 0:
        d[0::2] = s0 * c0
        d[1::2] = s1 * c1 * z1m0
    else:
        d[0::2] = s1 * c0
        d[1::2] = s0 * c1 * z1m1
    return d


def crc16(bits):
    reg = 0
    for b in bits:
        fb = ((reg >> 15) & 1) ^ int(b)
        reg = (reg << 1) & 0xFFFF
        if fb:
            reg ^= 0x1021
    return reg


def conv_encode_tb(bits):
    """Tail-biting rate-1/3 K=7 encoder (133,171,165): register initialised with the last 6 bits."""
    polys = [0o133, 0o171, 0o165]
    n = len(bits)
    state = 0
    for i in range(6):  # c_{-1} ... c_{-6} = last bits, newest at bit 5
        state |= int(bits[n - 1 - i]) << (5 - i)
    out = np.zeros((3, n), np.uint8)
    for k in range(n):
        reg = (int(bits[k]) << 6) | state
        for i, p in enumerate(polys):
            out[i, k] = bin(reg & p).count("1") & 1
        state = reg >> 1
    return out


PERM = [1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31,
        0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30]


def rate_match(d, e_len):
    D = d.shape[1]
    C = 32
    R = (D + C - 1) // C
    nd = R * C - D
    w = []
    for s in range(3):
        for col in PERM:
            for row in range(R):
                idx = row * C + col - nd
                w.append(-1 if idx < 0 else int(d[s, idx]))
    w = np.array(w)
    out = []
    j = 0
    while len(out) < e_len:
        v = w[j % len(w)]
        if v >= 0:
            out.append(v)
        j += 1
    return np.array(out, np.uint8)


def mib_bits(nof_prb=6, phich_ext=0, phich_r=2, sfn=0):
    bw = [6, 15, 25, 50, 75, 100].index(nof_prb)
    bits = [(bw >> 2) & 1, (bw >> 1) & 1, bw & 1, phich_ext, (phich_r >> 1) & 1, phich_r & 1]
    bits += [(sfn >> (9 - i)) & 1 for i in range(8)]  # 8 MSBs of the 10-bit SFN
    bits += [0] * 10
    return np.array(bits, np.uint8)


def symbol_start(l, cp_ext):
    if cp_ext:
        return l * (FFT + 32), 32
    return (0, 10) if l == 0 else (138 + (l - 1) * 137, 9)


def crs_k_and_values(cell_id, port, ns, l, cp_ext):
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
    k = 6 * np.arange(12) + (v + cell_id % 6) % 6
    n_cp = 0 if cp_ext else 1
    c_init = (1 << 10) * (7 * (ns + 1) + l + 1) * (2 * cell_id + 1) + 2 * cell_id + n_cp
    c = gold(c_init, 440).astype(float)
    m = np.arange(12) + 104
    r = ((1 - 2 * c[2 * m]) + 1j * (1 - 2 * c[2 * m + 1])) / np.sqrt(2)
    return k, r


def make_frame(cell_id, cp_ext=False, nof_ports=1, nof_prb=6, phich_ext=0, phich_r=2, sfn=0,
               channel=None, fill_data=True, seed=0, amplitude=None):
    """One 10 ms radio frame (19200 samples) as seen by a single receive antenna.
    channel: per-port complex gains (flat); default 1 for every port."""
    rng = np.random.default_rng(seed)
    nsymb = 6 if cp_ext else 7
    n_id_1, n_id_2 = cell_id // 3, cell_id % 3
    if channel is None:
        channel = np.ones(nof_ports, complex)
    channel = np.asarray(channel, complex)
    # per-port grids [port][slot][symbol][k]
    grid = np.zeros((nof_ports, 20, nsymb, NRE), complex)
    used = np.zeros((20, nsymb, NRE), bool)
    for ns in range(20):
        for l in range(nsymb):
            for p in range(nof_ports):
                kv = crs_k_and_values(cell_id, p, ns, l, cp_ext)
                if kv is not None:
                    grid[p, ns, l, kv[0]] = kv[1]
            # REs of CRS of any configured port are unused on the other ports
            for p in range(nof_ports):
                kv = crs_k_and_values(cell_id, p, ns, l, cp_ext)
                if kv is not None:
                    used[ns, l, kv[0]] = True
    # PSS / SSS on every port 0 (single-antenna-port transmission)
    for ns, sf in ((0, 0), (10, 5)):
        grid[:, ns, nsymb - 1, 5:67] = 0
        grid[0, ns, nsymb - 1, 5:67] = pss_seq(n_id_2)
        grid[:, ns, nsymb - 2, 5:67] = 0
        grid[0, ns, nsymb - 2, 5:67] = sss_seq(n_id_1, n_id_2, sf)
        used[ns, nsymb - 1, :] = True
        used[ns, nsymb - 2, :] = True
    # PBCH
    a = mib_bits(nof_prb, phich_ext, phich_r, sfn)
    crc = crc16(a) ^ {1: 0x0000, 2: 0xFFFF, 4: 0x5555}[nof_ports]
    c = np.concatenate([a, [(crc >> (15 - i)) & 1 for i in range(16)]]).astype(np.uint8)
    e_len = 1728 if cp_ext else 1920
    e = rate_match(conv_encode_tb(c), e_len)
    b = e ^ gold(cell_id, e_len)
    seg = e_len // 4
    frame_idx = sfn % 4
    bits = b[seg * frame_idx: seg * (frame_idx + 1)]
    d = ((1 - 2.0 * bits[0::2]) + 1j * (1 - 2.0 * bits[1::2])) / np.sqrt(2)
    # PBCH REs: slot 1, symbols 0..3, skipping the CRS positions of ports 0..3
    res = []
    for l in range(4):
        has_crs = l in ((0, 1, 3) if cp_ext else (0, 1))
        for k in range(NRE):
            if has_crs and (k % 3) == (cell_id % 3):
                continue
            res.append((l, k))
    assert len(res) == d.size
    if nof_ports == 1:
        for (l, k), v in zip(res, d):
            grid[0, 1, l, k] = v
    else:
        for i in range(0, d.size, 2):
            x0, x1 = d[i], d[i + 1]
            if nof_ports == 2:
                pa, pb = 0, 1
            else:
                pa, pb = (0, 2) if (i // 2) % 2 == 0 else (1, 3)
            (l0, k0), (l1, k1) = res[i], res[i + 1]
            grid[pa, 1, l0, k0] = x0 / np.sqrt(2)
            grid[pb, 1, l0, k0] = -np.conj(x1) / np.sqrt(2)
            grid[pa, 1, l1, k1] = x1 / np.sqrt(2)
            grid[pb, 1, l1, k1] = np.conj(x0) / np.sqrt(2)
    for (l, k) in res:
        used[1, l, k] = True
    for l in ((0, 1, 3) if cp_ext else (0, 1)):  # reserved for the CRS of ports 0..3: left empty
        used[1, l, (np.arange(NRE) % 3) == (cell_id % 3)] = True
    if fill_data:
        free = ~used
        n = int(free.sum())
        qpsk = ((1 - 2.0 * rng.integers(0, 2, n)) + 1j * (1 - 2.0 * rng.integers(0, 2, n))) / np.sqrt(2)
        g0 = grid[0]
        g0[free] = qpsk
    # OFDM modulation, ports summed through the flat channel
    x = np.zeros(19200, complex)
    for ns in range(20):
        for l in range(nsymb):
            f = np.tensordot(channel, grid[:, ns, l, :], axes=(0, 0))
            X = np.zeros(FFT, complex)
            X[FFT - 36:] = f[:36]
            X[1:37] = f[36:]
            t = np.fft.ifft(X) * FFT
            start, cp = symbol_start(l, cp_ext)
            s0 = ns * 960 + start
            x[s0:s0 + cp] = t[FFT - cp:]
            x[s0 + cp:s0 + cp + FFT] = t
    p = np.mean(np.abs(x) ** 2)
    if amplitude is None:
        amplitude = np.sqrt(0.5 / p)  # fixtures have mean |x|^2 ~ 0.5
    return (x * amplitude).astype(np.complex64)
