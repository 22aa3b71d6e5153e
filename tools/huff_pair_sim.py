"""Design aid for the entropy chain: how many symbols would a state-only walk (sync_span,
huffman_kernels.hip) consume per dependent table look-up if an entry covered TWO consecutive symbols whenever both fit the
look-up window?  Walks the true chain of bench-pool frames (640x480, q90, 4:2:0) and counts look-ups for window widths W.
Rules, as an implementation would need them to stay exact: a pair entry exists when symbol 1 is an AC symbol that does not
end the block (not EOB), its code + magnitude bits L1 and symbol 2's code + magnitude bits L2 (same AC table) satisfy
L1 + L2 <= W, and the zigzag index in front of the pair is < 48 (so symbol 1 cannot run past 63 unnoticed); with --dc also
DC symbol + first AC symbol.  Not on any product or test path."""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from infercam_onnx_amd import synth  # noqa: E402
sys.argv = sys.argv[:1]
from huff_sync_sim import build, parse  # noqa: E402


CODE_ONLY = True


def walk(jpeg, W, with_dc):
    dht, blocks, data = parse(jpeg)
    luts = {k: build(*v) for k, v in dht.items()}
    bits = int.from_bytes(data + b"\0" * 8, "big")
    nb = (len(data) + 8) * 8
    total = len(data) * 8
    bpm = len(blocks)

    def sym_at(p, lut):
        e = int(lut[(bits >> (nb - p - 16)) & 0xFFFF])
        ln, s = (e >> 8, e & 0xFF) if e else (1, 0)
        return ln, s

    p, c, z = 0, 0, 0
    nsym = nlook = npair = 0
    mcus = 0
    total_mcus = 40 * 30
    while mcus < total_mcus and p < total:
        td, ta = blocks[c]
        is_dc = z == 0
        ln, s = sym_at(p, luts[(0, td)] if is_dc else luts[(1, ta)])
        sz, run = s & 15, s >> 4
        L1 = ln + sz
        nlook += 1
        nsym += 1
        p += L1
        if is_dc:
            z = 1
            first_ok = with_dc
        else:
            first_ok = (sz != 0 or run == 15) and z < 48
            z = z + run + 1 if sz else (z + 16 if run == 15 else 64)
        if z < 64 and first_ok and L1 <= W:
            ln2, s2 = sym_at(p, luts[(1, ta)])
            sz2, run2 = s2 & 15, s2 >> 4
            if L1 + (ln2 if CODE_ONLY else ln2 + sz2) <= W:  # (state-only walks never look at magnitude bits)
                nsym += 1
                npair += 1
                p += ln2 + sz2
                z = z + run2 + 1 if sz2 else (z + 16 if run2 == 15 else 64)
        if z >= 64:
            z = 0
            c += 1
            if c == bpm:
                c = 0
                mcus += 1
    return nsym, nlook, npair


def main():
    frames = [synth.encode_jpeg(synth.synth_frame(synth.DEFAULT_FRAME_SEED, i, 640, 480), quality=90, subsampling="4:2:0") for i in range(0, 64, 8)]
    for with_dc in (False, True):
        for W in (10, 11, 12, 13, 14, 16):
            ns = nl = 0
            for j in frames:
                a, b, _ = walk(j, W, with_dc)
                ns += a
                nl += b
            print("pairs: AC+AC%s  window %2d bits: %.3f symbols per look-up (%d symbols, %d look-ups over %d frames)" %
                  (" and DC+AC" if with_dc else "          ", W, ns / nl, ns, nl, len(frames)))


if __name__ == "__main__":
    main()
