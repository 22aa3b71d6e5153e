"""Host simulation of the self-synchronising subsequence decoder (huffman_kernels.hip the k_huff_* pipeline):
how many fixed-point rounds does a frame need, and how many subsequences are re-decoded per round?
Design aid only -- not on any product or test path."""
import sys
import numpy as np
from infercam_onnx_amd import synth


def parse(jpeg):
    p = 2
    dht = {}
    comps = []
    while True:
        assert jpeg[p] == 0xFF
        m = jpeg[p + 1]
        L = (jpeg[p + 2] << 8) | jpeg[p + 3]
        seg = jpeg[p + 4:p + 2 + L]
        if m == 0xC4:
            q = 0
            while q < len(seg):
                tc, th = seg[q] >> 4, seg[q] & 15
                counts = list(seg[q + 1:q + 17])
                n = sum(counts)
                syms = list(seg[q + 17:q + 17 + n])
                dht[(tc, th)] = (counts, syms)
                q += 17 + n
        elif m == 0xC0:
            nc = seg[5]
            for i in range(nc):
                comps.append((seg[6 + 3 * i], seg[7 + 3 * i] >> 4, seg[7 + 3 * i] & 15))
        elif m == 0xDA:
            ns = seg[0]
            sel = {seg[1 + 2 * i]: (seg[2 + 2 * i] >> 4, seg[2 + 2 * i] & 15) for i in range(ns)}
            data = jpeg[p + 2 + L:]
            break
        p += 2 + L
    # unstuff
    out = bytearray()
    i = 0
    while i < len(data):
        b = data[i]
        if b == 0xFF:
            if data[i + 1] == 0:
                out.append(0xFF)
                i += 2
                continue
            break
        out.append(b)
        i += 1
    blocks = []
    for cid, h, v in comps:
        td, ta = sel[cid]
        blocks += [(td, ta)] * (h * v)
    return dht, blocks, bytes(out)


def build(counts, syms):
    """16-bit lookup: code prefix -> (len, sym)"""
    lut = np.zeros(65536, dtype=np.int32)
    code = 0
    k = 0
    for l in range(1, 17):
        for _ in range(counts[l - 1]):
            lo = code << (16 - l)
            lut[lo:lo + (1 << (16 - l))] = (l << 8) | syms[k]
            k += 1
            code += 1
        code <<= 1
    return lut


def main():
    W, H = 640, 480
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    sub_bytes = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    jpeg = synth.encode_jpeg(synth.synth_frame(0x5EED0000, seed, W, H), quality=90, subsampling="4:2:0")
    dht, blocks, data = parse(jpeg)
    luts = {k: build(*v) for k, v in dht.items()}
    bits = int.from_bytes(data + b"\0" * 8, "big")
    nbits_total = (len(data) + 8) * 8
    total_bits = len(data) * 8
    bpm = len(blocks)

    def peek16(pos):
        return (bits >> (nbits_total - pos - 16)) & 0xFFFF

    def span(p, c, z, limit):
        nsym = 0
        while p < limit:
            td, ta = blocks[c]
            lut = luts[(0, td)] if z == 0 else luts[(1, ta)]
            e = int(lut[peek16(p)])
            if e == 0:
                ln, sym = 1, 0
            else:
                ln, sym = e >> 8, e & 0xFF
            sz, run = sym & 15, sym >> 4
            p += ln + sz
            nsym += 1
            if z == 0:
                z = 1
            elif sz:
                z = z + run + 1
            else:
                z = z + 16 if run == 15 else 64
            if z >= 64:
                z = 0
                c = (c + 1) % bpm
        return (p, c, z), nsym

    sub_bits = sub_bytes * 8
    nsub = (len(data) + sub_bytes - 1) // sub_bytes
    if len(sys.argv) > 3:
        multi(span, nsub, sub_bits, total_bits, bpm, sys.argv[3])
        return
    S = [(i * sub_bits, 0, 0) for i in range(nsub + 1)]
    last = [None] * nsub
    ex = [None] * nsub
    rounds = 0
    print("bytes", len(data), "nsub", nsub, "bpm", bpm)
    while True:
        redo = [i for i in range(nsub) if S[i] != last[i]]
        for i in redo:
            ex[i], _ = span(*S[i], min((i + 1) * sub_bits, total_bits))
            last[i] = S[i]
        changed = 0
        kinds = {"p": 0, "c_only": 0, "z": 0}
        for i in redo:
            if S[i + 1] != ex[i]:
                a, b = S[i + 1], ex[i]
                if a[0] != b[0]:
                    kinds["p"] += 1
                elif a[2] != b[2]:
                    kinds["z"] += 1
                else:
                    kinds["c_only"] += 1
                S[i + 1] = ex[i]
                changed += 1
        rounds += 1
        print("round", rounds, "redo", len(redo), "changed", changed, kinds)
        if not changed:
            break


def multi(span, nsub, sub_bits, total_bits, bpm, mode):
    """Several hypotheses per boundary; every round decodes each not-yet-cached candidate entry."""
    if mode == "c":
        seeds = lambda i: [(i * sub_bits, g, 0) for g in range(bpm)]
    elif mode == "cz":
        seeds = lambda i: [(i * sub_bits, g, z) for g in range(bpm) for z in (0, 1)]
    else:
        seeds = lambda i: [(i * sub_bits, 0, 0)]
    cache = [dict() for _ in range(nsub)]
    cand = [set(seeds(i)) for i in range(nsub + 1)]
    cand[0] = {(0, 0, 0)}
    true_i, true_s = 0, (0, 0, 0)
    rounds = 0
    while true_i < nsub:
        work = 0
        maxw = 0
        new = [set() for _ in range(nsub + 1)]
        for i in range(nsub):
            todo = [e for e in cand[i] if e not in cache[i]]
            maxw = max(maxw, len(todo))
            for e in todo:
                cache[i][e], _ = span(*e, min((i + 1) * sub_bits, total_bits))
                new[i + 1].add(cache[i][e])
                work += 1
        for i in range(nsub + 1):
            cand[i] |= new[i]
        while true_i < nsub and true_s in cache[true_i]:
            true_s = cache[true_i][true_s]
            true_i += 1
        rounds += 1
        print("round", rounds, "decodes", work, "max per sub", maxw, "truth reached", true_i, "of", nsub)


if __name__ == "__main__":
    main()
