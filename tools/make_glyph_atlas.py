#!/usr/bin/env python3
"""Glyph coverage atlas for the confidence label of row N1 (inferer.rs:80-88):

    draw_text(&frame, color, x_tl as i32, y_tl as i32, Scale { x: 16.0, y: 16.0 }, &DEJAVU_MONO, &format!("{:.2}%", confidence * 100.0))

The reference rasterises the label with rusttype 0.9.3 (+ owned_ttf_parser / ttf-parser for the outlines and
ab_glyph_rasterizer for the coverage) and imageproc 0.23 blends it in; none of those crates is under /root/reference
or in this image, so this script RESTATES their published algorithms (float32 arithmetic step by step) and emits the
result as DATA: for every character position k of a label (the caret advances by a fractional amount, so the sub-pixel
phase differs per position) and every glyph of "0123456789.%", the pixel bounding box relative to the text origin and the
per-pixel coverage.  The product (csrc/glyph_atlas.inc) and the oracle (oracle/glyph_atlas.inc) only blend these tables.
"Parity unpinned": there is nothing here to run rusttype against; FreeType (PIL) is used as a sanity check of the
shapes only (tests/test_oracle_encode.py).

Input: the reference's resources/DejaVuSansMono.ttf (read as a data file when present), else the system's DejaVuSansMono.
Run in the build container:  python tools/make_glyph_atlas.py
"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FONT_CANDIDATES = ["/root/reference/resources/DejaVuSansMono.ttf", "/usr/share/fonts/truetype/dejavu/DejaVuSansMono.ttf"]
CHARS = "0123456789.%"
MAX_LEN = 8          # "100.00%" has 7 characters
f32 = np.float32


class Font:
    def __init__(self, path):
        self.d = d = open(path, "rb").read()
        n = struct.unpack(">H", d[4:6])[0]
        self.t = {}
        for i in range(n):
            tag, _, off, ln = struct.unpack(">4sIII", d[12 + 16 * i:28 + 16 * i])
            self.t[tag.decode()] = (off, ln)
        o = self.t["head"][0]
        self.loc_long = struct.unpack(">h", d[o + 50:o + 52])[0]
        o = self.t["hhea"][0]
        self.ascender, self.descender = struct.unpack(">hh", d[o + 4:o + 8])
        self.n_hmetrics = struct.unpack(">H", d[o + 34:o + 36])[0]
        assert "kern" not in self.t  # rusttype's pair_kerning reads the kern table only: none here

    def glyph_id(self, ch):
        """cmap format 4 (Unicode BMP)."""
        d = self.d
        o = self.t["cmap"][0]
        n = struct.unpack(">H", d[o + 2:o + 4])[0]
        sub = None
        for i in range(n):
            pid, eid, off = struct.unpack(">HHI", d[o + 4 + 8 * i:o + 12 + 8 * i])
            if (pid, eid) in ((3, 1), (0, 3)) and struct.unpack(">H", d[o + off:o + off + 2])[0] == 4:
                sub = o + off
        assert sub is not None
        segx2 = struct.unpack(">H", d[sub + 6:sub + 8])[0]
        seg = segx2 // 2
        ends = struct.unpack(">%dH" % seg, d[sub + 14:sub + 14 + segx2])
        starts = struct.unpack(">%dH" % seg, d[sub + 16 + segx2:sub + 16 + 2 * segx2])
        deltas = struct.unpack(">%dh" % seg, d[sub + 16 + 2 * segx2:sub + 16 + 3 * segx2])
        ro_base = sub + 16 + 3 * segx2
        ros = struct.unpack(">%dH" % seg, d[ro_base:ro_base + segx2])
        c = ord(ch)
        for i in range(seg):
            if starts[i] <= c <= ends[i]:
                if ros[i] == 0:
                    return (c + deltas[i]) & 0xFFFF
                p = ro_base + 2 * i + ros[i] + 2 * (c - starts[i])
                g = struct.unpack(">H", d[p:p + 2])[0]
                return (g + deltas[i]) & 0xFFFF if g else 0
        return 0

    def advance(self, gid):
        o = self.t["hmtx"][0]
        i = min(gid, self.n_hmetrics - 1)
        return struct.unpack(">H", self.d[o + 4 * i:o + 4 * i + 2])[0]

    def _glyph_data(self, gid):
        o = self.t["loca"][0]
        if self.loc_long:
            a, b = struct.unpack(">II", self.d[o + 4 * gid:o + 4 * gid + 8])
        else:
            a, b = (2 * v for v in struct.unpack(">HH", self.d[o + 2 * gid:o + 2 * gid + 4]))
        g = self.t["glyf"][0]
        return self.d[g + a:g + b]

    def contours(self, gid):
        """Simple glyph -> list of contours, each a list of (x, y, on_curve) in font units."""
        d = self._glyph_data(gid)
        nc = struct.unpack(">h", d[0:2])[0]
        assert nc >= 0, "composite glyph: not needed for the label's characters"
        ends = struct.unpack(">%dH" % nc, d[10:10 + 2 * nc])
        p = 10 + 2 * nc
        ilen = struct.unpack(">H", d[p:p + 2])[0]
        p += 2 + ilen
        npts = ends[-1] + 1
        flags = []
        while len(flags) < npts:
            f = d[p]
            p += 1
            flags.append(f)
            if f & 8:
                r = d[p]
                p += 1
                flags.extend([f] * r)
        xs, ys = [], []
        v = 0
        for f in flags:
            if f & 2:
                dx = d[p]
                p += 1
                v += dx if f & 16 else -dx
            elif not f & 16:
                v += struct.unpack(">h", d[p:p + 2])[0]
                p += 2
            xs.append(v)
        v = 0
        for f in flags:
            if f & 4:
                dy = d[p]
                p += 1
                v += dy if f & 32 else -dy
            elif not f & 32:
                v += struct.unpack(">h", d[p:p + 2])[0]
                p += 2
            ys.append(v)
        out, s = [], 0
        for e in ends:
            out.append([(xs[i], ys[i], bool(flags[i] & 1)) for i in range(s, e + 1)])
            s = e + 1
        return out


def outline(contours, emit):
    """ttf-parser's glyf outliner: on/off-curve points -> move_to / line_to / quad_to / close (font units, f32)."""
    def lerp(a, b, t):
        return (f32(a[0] + f32(t) * f32(b[0] - a[0])), f32(a[1] + f32(t) * f32(b[1] - a[1])))

    for pts in contours:
        first_on = first_off = last_off = None
        for (x, y, on) in pts:
            p = (f32(x), f32(y))
            if first_on is None:
                if on:
                    first_on = p
                    emit("M", p)
                elif first_off is not None:
                    mid = lerp(first_off, p, 0.5)
                    first_on, last_off = mid, p
                    emit("M", mid)
                else:
                    first_off = p
            else:
                if last_off is not None and on:
                    emit("Q", last_off, p)
                    last_off = None
                elif last_off is not None:
                    mid = lerp(last_off, p, 0.5)
                    emit("Q", last_off, mid)
                    last_off = p
                elif on:
                    emit("L", p)
                else:
                    last_off = p
        if first_off is not None and last_off is not None:
            mid = lerp(last_off, first_off, 0.5)
            emit("Q", last_off, mid)
            last_off = None
        if first_on is not None and first_off is not None:
            emit("Q", first_off, first_on)
        elif first_on is not None and last_off is not None:
            emit("Q", last_off, first_on)
        elif first_on is not None:
            emit("L", first_on)
        emit("Z")


class Rasterizer:
    """ab_glyph_rasterizer 0.1: signed-area accumulation, float32 throughout."""

    def __init__(self, w, h):
        self.w, self.h = w, h
        self.a = np.zeros(w * h + 4, np.float32)

    def draw_line(self, p0, p1):
        if abs(f32(p0[1] - p1[1])) <= np.finfo(np.float32).eps:
            return
        if p0[1] < p1[1]:
            d_, a0_, a1_ = f32(1.0), p0, p1
        else:
            d_, a0_, a1_ = f32(-1.0), p1, p0
        p0, p1 = a0_, a1_
        dxdy = f32(f32(p1[0] - p0[0]) / f32(p1[1] - p0[1]))
        x = f32(p0[0])
        y0 = max(int(p0[1]), 0) if p0[1] >= 0 else 0  # `as usize`: saturating, truncating
        if p0[1] < 0:
            x = f32(x - f32(p0[1] * dxdy))
        yend = min(self.h, max(int(np.ceil(p1[1])), 0))
        for y in range(y0, yend):
            linestart = y * self.w
            dy = f32(min(f32(y + 1), p1[1]) - max(f32(y), p0[1]))
            xnext = f32(x + f32(dxdy * dy))
            d = f32(dy * d_)
            x0, x1 = (x, xnext) if x < xnext else (xnext, x)
            x0floor = f32(np.floor(x0))
            x0i = int(x0floor)
            x1ceil = f32(np.ceil(x1))
            x1i = int(x1ceil)
            if x1i <= x0i + 1:
                xmf = f32(f32(f32(0.5) * f32(x + xnext)) - x0floor)
                i = linestart + x0i
                if i < 0:
                    x = xnext
                    continue
                self.a[i] = f32(self.a[i] + f32(d - f32(d * xmf)))
                self.a[i + 1] = f32(self.a[i + 1] + f32(d * xmf))
            else:
                s = f32(f32(1.0) / f32(x1 - x0))
                x0f = f32(x0 - x0floor)
                om = f32(f32(1.0) - x0f)
                a0 = f32(f32(f32(f32(0.5) * s) * om) * om)
                x1f = f32(f32(x1 - x1ceil) + f32(1.0))
                am = f32(f32(f32(f32(0.5) * s) * x1f) * x1f)
                i = linestart + x0i
                if i < 0:
                    x = xnext
                    continue
                self.a[i] = f32(self.a[i] + f32(d * a0))
                if x1i == x0i + 2:
                    self.a[i + 1] = f32(self.a[i + 1] + f32(d * f32(f32(f32(1.0) - a0) - am)))
                else:
                    a1 = f32(s * f32(f32(1.5) - x0f))
                    self.a[i + 1] = f32(self.a[i + 1] + f32(d * f32(a1 - a0)))
                    for xi in range(x0i + 2, x1i - 1):
                        self.a[linestart + xi] = f32(self.a[linestart + xi] + f32(d * s))
                    a2 = f32(a1 + f32(f32(x1i - x0i - 3) * s))
                    self.a[linestart + x1i - 1] = f32(self.a[linestart + x1i - 1] + f32(d * f32(f32(f32(1.0) - a2) - am)))
                self.a[linestart + x1i] = f32(self.a[linestart + x1i] + f32(d * am))
            x = xnext

    def draw_quad(self, p0, p1, p2):
        devx = f32(f32(p0[0] - f32(f32(2.0) * p1[0])) + p2[0])
        devy = f32(f32(p0[1] - f32(f32(2.0) * p1[1])) + p2[1])
        devsq = f32(f32(devx * devx) + f32(devy * devy))
        if devsq < f32(0.333):
            self.draw_line(p0, p2)
            return
        n = 1 + int(np.floor(np.sqrt(np.sqrt(f32(f32(3.0) * devsq), dtype=np.float32), dtype=np.float32)))
        p = p0
        nrecip = f32(f32(1.0) / f32(n))
        t = f32(0.0)

        def lerp(t, a, b):
            return (f32(a[0] + f32(t * f32(b[0] - a[0]))), f32(a[1] + f32(t * f32(b[1] - a[1]))))

        for _ in range(n - 1):
            t = f32(t + nrecip)
            pn = lerp(t, lerp(t, p0, p1), lerp(t, p1, p2))
            self.draw_line(p, pn)
            p = pn
        self.draw_line(p, p2)

    def coverage(self):
        out = np.zeros(self.w * self.h, np.float32)
        acc = f32(0.0)
        for i in range(self.w * self.h):
            acc = f32(acc + self.a[i])
            out[i] = min(abs(acc), f32(1.0))
        return out.reshape(self.h, self.w)


def build_atlas(font_path):
    font = Font(font_path)
    px = f32(16.0)
    fheight = f32(f32(font.ascender) - f32(font.descender))
    scale_y = f32(px / fheight)                      # Font::scale_for_pixel_height
    scale_x = f32(f32(scale_y * px) / px)            # ScaledGlyph: scale_y * scale.x / scale.y
    ascent = f32(f32(font.ascender) * scale_y)       # v_metrics(scale).ascent: the layout's start.y
    glyphs = {}
    for ch in CHARS:
        gid = font.glyph_id(ch)
        cs = font.contours(gid)
        segs = []
        outline(cs, lambda *a: segs.append(a))
        pts = [p for s in segs for p in s[1:]]
        bbox = (min(p[0] for p in pts), min(p[1] for p in pts), max(p[0] for p in pts), max(p[1] for p in pts))
        glyphs[ch] = (gid, segs, bbox, font.advance(gid))
    atlas = {}
    advs = {g[3] for g in glyphs.values()}
    assert len(advs) == 1, "monospace: one advance"  # the caret position then depends on the character index only
    adv = f32(f32(advs.pop()) * scale_x)
    caret = f32(0.0)
    for k in range(MAX_LEN):
        pos = (f32(f32(0.0) + caret), ascent)  # point(start.x + caret, start.y)
        for ch, (gid, segs, (xmin, ymin, xmax, ymax), _) in glyphs.items():
            bx0 = int(np.floor(f32(f32(f32(xmin) * scale_x) + pos[0])))
            by0 = int(np.floor(f32(f32(f32(-ymax) * scale_y) + pos[1])))
            bx1 = int(np.ceil(f32(f32(f32(xmax) * scale_x) + pos[0])))
            by1 = int(np.ceil(f32(f32(f32(-ymin) * scale_y) + pos[1])))
            w, h = bx1 - bx0, by1 - by0
            ras = Rasterizer(w, h)
            tr = (f32(pos[0] - f32(bx0)), f32(pos[1] - f32(by0)))  # OutlineTranslator(position - bb.min)
            nsy = f32(-scale_y)

            def xf(p):  # OutlineScaler then OutlineTranslator
                return (f32(f32(p[0] * scale_x) + tr[0]), f32(f32(p[1] * nsy) + tr[1]))

            last = last_move = None
            for s in segs:
                if s[0] == "M":
                    last = last_move = xf(s[1])
                elif s[0] == "L":
                    e = xf(s[1])
                    ras.draw_line(last, e)
                    last = e
                elif s[0] == "Q":
                    c, e = xf(s[1]), xf(s[2])
                    ras.draw_quad(last, c, e)
                    last = e
                else:
                    if last_move is not None:
                        ras.draw_line(last, last_move)
            atlas[(k, ch)] = (bx0, by0, w, h, ras.coverage())
        caret = f32(caret + adv)
    return atlas


def write_inc(atlas, path):
    lines = ["// GENERATED by tools/make_glyph_atlas.py -- data, do not edit.  Coverage of the label glyphs of inferer.rs:80-88",
             "// (DejaVuSansMono, 16 px) per character position k (sub-pixel phase of the caret) and glyph of \"0123456789.%\":",
             "// bounding box relative to the text origin (x_tl as i32, y_tl as i32) and row-major f32 coverage.",
             "#define UFD_GLYPH_POSITIONS %d" % MAX_LEN, "#define UFD_GLYPH_CHARS %d" % len(CHARS),
             "typedef struct UfdGlyph { int x, y, w, h, off; } UfdGlyph;", "static const UfdGlyph kUfdGlyphs[UFD_GLYPH_POSITIONS][UFD_GLYPH_CHARS] = {"]
    cov, off = [], 0
    for k in range(MAX_LEN):
        row = []
        for ch in CHARS:
            x, y, w, h, c = atlas[(k, ch)]
            row.append("{%d, %d, %d, %d, %d}" % (x, y, w, h, off))
            cov.append(c.ravel())
            off += w * h
        lines.append("  {" + ", ".join(row) + "},")
    lines.append("};")
    flat = np.concatenate(cov)
    lines.append("#define UFD_GLYPH_COVERAGE_FLOATS %d" % flat.size)
    lines.append("static const float kUfdGlyphCoverage[UFD_GLYPH_COVERAGE_FLOATS] = {")
    for i in range(0, flat.size, 8):
        lines.append("  " + ", ".join(("%.9gf" % v) if v not in (0.0, 1.0) else ("%.1ff" % v) for v in flat[i:i + 8]) + ",")
    lines.append("};")
    open(path, "w").write("\n".join(lines) + "\n")


def main():
    font = next(p for p in FONT_CANDIDATES if os.path.exists(p))
    atlas = build_atlas(font)
    for rel in ("infercam_onnx_amd/csrc/glyph_atlas.inc", "oracle/glyph_atlas.inc"):
        write_inc(atlas, os.path.join(ROOT, rel))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "glyph_atlas.npz"),
                        **{"%d_%s" % (k, "pct" if ch == "%" else ("dot" if ch == "." else ch)): np.concatenate(
                            [np.array([x, y, w, h], np.float32), c.ravel()]) for (k, ch), (x, y, w, h, c) in atlas.items()})
    x, y, w, h, c = atlas[(0, "8")]
    print("font", font, "| glyph '8' at position 0: bb (%d,%d) %dx%d" % (x, y, w, h))
    for r in c:
        print("".join(" .:-=+*#%@"[min(9, int(v * 9.999))] for v in r))


if __name__ == "__main__":
    main()
