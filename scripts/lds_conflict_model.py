"""CPU model of the LDS bank conflicts of the p = 3 streaming kernel's contraction passes (pa_nd_hex_core.hpp,
pa_nd_hex_stream.hip), with the banking rules of MI355X_MICROARCH.md (LDS section):
  ds_read_b64   two groups of 32 lanes, bank = (byte address / 4) mod 64
  ds_write_b64  four groups of 16 lanes, bank = (byte address / 4) mod 32
A group costs one LDS cycle plus one per extra distinct address on its busiest bank.  Prints, per element stride (doubles),
the extra cycles per batch of the swizzled layouts.  The kernel places element `sub` of a wave at sub * LDS_ELEM doubles and
flips bit 4 of the in-element index for odd `sub` (parity_xor): that assumes LDS_ELEM = 0 (mod 32 doubles)."""
import sys


def group_cost(addrs, banks):
    """addrs: double indices of the lanes of one group (None = inactive).  8-byte accesses: double-bank = index mod (banks / 2)."""
    per = {}
    for a in addrs:
        if a is None:
            continue
        per.setdefault(a % (banks // 2), set()).add(a)
    return max([len(v) for v in per.values()] + [1])


def ia(f, qx, j, k):
    return f * 64 + qx * 16 + (((j ^ qx) & 3) << 2) + ((k ^ qx) & 3)


def patterns(P1=3, inplace=True):
    """(kind, fn(ta, tb) -> in-element double index or None) for every LDS access of the three components, curl-curl."""
    NC, Q1 = P1 + 1, 4
    boff = 0 if inplace else 128
    ib = lambda f, qx, qy, k: boff + f * 64 + qx * 16 + (((qy ^ qx) & 3) << 2) + ((k ^ qx) & 3)  # noqa: E731
    out = []
    for C in range(3):
        ni, nj, nk = (P1 if C == 0 else NC), (P1 if C == 1 else NC), (P1 if C == 2 else NC)
        DX, DY = C != 0, C != 1
        nA = 2 if DX else 1
        nB = 1 + (1 if DY else 0) + (1 if DX else 0)
        # E staging read (uin): sm[C*P1*NC*NC + i + ni*(ta + nj*tb)]
        for i in range(ni):
            out.append(("r", lambda ta, tb, i=i, C=C, ni=ni, nj=nj, nk=nk: (C * P1 * NC * NC + i + ni * (ta + nj * tb)) if (ta < nj and tb < nk) else None, False))
        # fwd X write: ia(f, qx, ta, tb)
        for qx in range(Q1):
            for f in range(nA):
                out.append(("w", lambda ta, tb, f=f, qx=qx, nj=nj, nk=nk: ia(f, qx, ta, tb) if (ta < nj and tb < nk) else None, True))
        # fwd Y read ia(f, ta, j, tb)
        for j in range(nj):
            for f in range(nA):
                out.append(("r", lambda ta, tb, f=f, j=j, nk=nk: ia(f, ta, j, tb if tb < nk else 0), True))
        for qy in range(Q1):
            for f in range(nB):
                out.append(("w", lambda ta, tb, f=f, qy=qy, nk=nk: ib(f, ta, qy, tb) if tb < nk else None, True))
        for k in range(nk):
            for f in range(nB):
                out.append(("r", lambda ta, tb, f=f, k=k: ib(f, ta, tb, k), True))
        # bwd Z^T write ib(f, ta, tb, k)
        for k in range(nk):
            for f in range(nB):
                out.append(("w", lambda ta, tb, f=f, k=k: ib(f, ta, tb, k), True))
        for qy in range(Q1):
            for f in range(nB):
                out.append(("r", lambda ta, tb, f=f, qy=qy, nk=nk: ib(f, ta, qy, tb if tb < nk else 0), True))
        for j in range(nj):
            for f in range(nA):
                out.append(("w", lambda ta, tb, f=f, j=j, nk=nk: ia(f, ta, j, tb) if tb < nk else None, True))
        for qx in range(Q1):
            for f in range(nA):
                out.append(("r", lambda ta, tb, f=f, qx=qx, nj=nj, nk=nk: ia(f, qx, ta if (ta < nj and tb < nk) else 0, tb if (ta < nj and tb < nk) else 0), True))
        # E^T staging write
        for i in range(ni):
            out.append(("w", lambda ta, tb, i=i, C=C, ni=ni, nj=nj, nk=nk: (C * P1 * NC * NC + i + ni * (ta + nj * tb)) if (ta < nj and tb < nk) else None, False))
    return out


def batch_extra_cycles(stride, use_xor, inplace=True):
    extra_r = extra_w = n_r = n_w = 0
    for kind, fn, swz in patterns(3, inplace):
        lanes = []
        for lane in range(64):
            sub, t = lane >> 4, lane & 15
            ta, tb = t & 3, t >> 2
            a = fn(ta, tb)
            if a is None:
                lanes.append(None)
                continue
            if swz and use_xor and (sub & 1):
                a ^= 16
            lanes.append(sub * stride + a)
        if kind == "r":
            n_r += 1
            extra_r += sum(group_cost(lanes[g * 32:(g + 1) * 32], 64) - 1 for g in range(2))
        else:
            n_w += 1
            extra_w += sum(group_cost(lanes[g * 16:(g + 1) * 16], 32) - 1 for g in range(4))
    return n_r, extra_r, n_w, extra_w


if __name__ == "__main__":
    print("stride xor  reads extra_read_cycles  writes extra_write_cycles   (curl-curl p=3, in-place swizzled layout)")
    for stride, x in ((300, True), (320, True), (304, False), (304, True), (288, True), (272, False), (336, False)):
        print(stride, x, *batch_extra_cycles(stride, x))
    print("K+M-style separate buffers (NDLayout<3,4>):")
    for stride, x in ((428, True), (448, True), (432, False)):
        print(stride, x, *batch_extra_cycles(stride, x, inplace=False))
