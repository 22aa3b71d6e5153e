// plan.cpp -- the network side of the handle: which of the 52 convolutions of UltraFace-RFB (what tract's
// SimplePlan::run executes for `self.model.run`, infer_server/src/nn.rs:181) run fused, chained, merged, stacked or riding in
// another launch's grid, where every activation tensor lives in the arena and for how long, how the weights are packed
// for the kernels, and the launch of every layer.  plan_tensors makes no HIP call: ufd_debug_plan exposes it to the CPU
// test suite.
#include "model_types.hpp"
#include <algorithm>
#include <cstdlib>
#include <vector>

using namespace ufd;

namespace ufd {

// ---------------------------------------------------------------- model construction
void gen_priors(int W, int H, std::vector<float>& out) {
  // upstream generate_priors: float64 arithmetic, cast to f32, clamp to [0, 1]
  out.clear();
  for (int idx = 0; idx < 4; idx++) {
    int fw = (W + kStrides[idx] - 1) / kStrides[idx], fh = (H + kStrides[idx] - 1) / kStrides[idx];
    double shrink_w = (double)W / fw, shrink_h = (double)H / fh;
    double scale_w = (double)W / shrink_w, scale_h = (double)H / shrink_h;
    for (int j = 0; j < fh; j++)
      for (int i = 0; i < fw; i++) {
        double xc = (i + 0.5) / scale_w, yc = (j + 0.5) / scale_h;
        for (int a = 0; a < kHeadAnchors[idx]; a++) {
          double v[4] = {xc, yc, kMinBoxes[idx][a] / W, kMinBoxes[idx][a] / H};
          for (double x : v) {
            float f = (float)x;
            out.push_back(f < 0.f ? 0.f : (f > 1.f ? 1.f : f));
          }
        }
      }
  }
}

// Liveness-based arena: every conv output gets [B][c][h][w]; buffers are recycled after their
// last reader unless UFD_FLAG_KEEP_LAYERS asks to keep all of them for ufd_debug_layer_output.
void plan_tensors(ufd_model* m, bool keep_all) {
  const uint32_t flags = m->cfg.flags;
  const ConvSpec* specs = conv_specs();
  m->layers.resize(kNumConv);
  m->tensors.clear();
  std::vector<int> tensor_of(kNumConv, -1);
  int cat_tensor = -1;
  for (int i = 0; i < kNumConv; i++) {
    Layer& L = m->layers[i];
    L.spec = specs[i];
    if (L.spec.src == -1) {
      L.ih = m->H, L.iw = m->W;
      L.in_tensor = -1;
    } else if (L.spec.src == -2) {
      L.ih = m->layers[kRfbCatA].oh, L.iw = m->layers[kRfbCatA].ow;
      L.in_tensor = cat_tensor;
    } else {
      L.ih = m->layers[L.spec.src].oh, L.iw = m->layers[L.spec.src].ow;
      L.in_tensor = tensor_of[L.spec.src];
    }
    L.oh = conv_out_dim(L.ih, L.spec);
    L.ow = conv_out_dim(L.iw, L.spec);
    if (L.spec.k == 1 && L.spec.groups == 1)
      L.kind = kKindPointwise;
    else if (L.spec.k == 3 && L.spec.groups == 1 && L.spec.cout <= 16 && L.spec.pad == L.spec.dil)
      L.kind = kKindConv3x3;
    else
      L.kind = kKindDirect;
    L.res_tensor = (i == kRfbShortcut) ? tensor_of[kRfbLinear] : -1;
    L.out_coff = 0;
    if (i == kRfbCatA || i == kRfbCatB || i == kRfbCatC) {
      if (cat_tensor < 0) {
        Tensor t;
        t.c = 48, t.h = L.oh, t.w = L.ow;
        m->tensors.push_back(t);
        cat_tensor = (int)m->tensors.size() - 1;
      }
      L.out_tensor = cat_tensor;
      L.out_coff = (i == kRfbCatA) ? 0 : (i == kRfbCatB ? 16 : 32);
    } else {
      Tensor t;
      t.c = L.spec.cout, t.h = L.oh, t.w = L.ow;
      m->tensors.push_back(t);
      L.out_tensor = (int)m->tensors.size() - 1;
    }
    tensor_of[i] = L.out_tensor;
    const double in_b = (double)L.spec.cin * L.ih * L.iw * 4, out_b = (double)L.spec.cout * L.oh * L.ow * 4;
    L.weight_bytes = (double)(conv_weight_floats(L.spec) + L.spec.cout) * 4;
    L.bytes_per_frame = in_b + out_b + (L.res_tensor >= 0 ? out_b : 0);
    L.flops_per_frame = 2.0 * L.oh * L.ow * L.spec.cout * (L.spec.cin / L.spec.groups) * L.spec.k * L.spec.k;
  }
  // fuse every depthwise 3x3 into the pointwise conv that consumes it (its only consumer)
  for (int i = 0; i + 1 < kNumConv; i++) {
    Layer& D = m->layers[i];
    Layer& P = m->layers[i + 1];
    if (D.spec.groups == 1 || D.spec.groups != D.spec.cin || D.spec.k != 3 || D.spec.pad != 1 || D.spec.dil != 1 ||
        !D.spec.relu)
      continue;
    if (P.kind != kKindPointwise || P.spec.src != i || P.res_tensor >= 0) continue;
    ConvArgs probe{};
    probe.ih = D.ih, probe.iw = D.iw, probe.oh = P.oh, probe.ow = P.ow, probe.cin = D.spec.cin;
    if (!dwpw_supported(probe, D.spec.stride)) continue;
    P.kind = kKindDwPw;
    P.fused_dw = i;
    D.kind = kKindFusedAway;
    D.materialize = keep_all;
    P.bytes_per_frame = (double)D.spec.cin * D.ih * D.iw * 4 + (double)P.spec.cout * P.oh * P.ow * 4;
    P.flops_per_frame += D.flops_per_frame;
    P.weight_bytes += D.weight_bytes;
  }
  // chain two dw->pw blocks into one launch where the tensor between them is the big one
  // (m1 -> m2: 32 channels at half the input resolution) and nothing else reads it
  if (!keep_all && !(flags & UFD_FLAG_NO_CHAIN)) {
    for (int i = 0; i < kNumConv; i++) {
      Layer& P2 = m->layers[i];
      if (P2.kind != kKindDwPw) continue;
      const Layer& D2 = m->layers[P2.fused_dw];
      const int p1 = D2.spec.src;
      if (p1 < 0 || m->layers[p1].kind != kKindDwPw || m->layers[p1].chained) continue;
      Layer& P1 = m->layers[p1];
      const Layer& D1 = m->layers[P1.fused_dw];
      bool only_reader = true;
      for (int j = 0; j < kNumConv; j++)
        if (j != P2.fused_dw && m->layers[j].spec.src == p1) only_reader = false;
      for (int h = 0; h < 4; h++)
        if (kHeadCls[h] == p1 || kHeadReg[h] == p1) only_reader = false;
      if (!only_reader || D1.spec.stride != 1 || D2.spec.stride != 2) continue;
      ConvArgs f{}, g{};
      f.cin = P1.spec.cin, f.cout = P1.spec.cout, f.ih = D1.ih, f.iw = D1.iw, f.oh = P1.oh, f.ow = P1.ow, f.relu = P1.spec.relu;
      g.cin = P2.spec.cin, g.cout = P2.spec.cout, g.ih = D2.ih, g.iw = D2.iw, g.oh = P2.oh, g.ow = P2.ow;
      if (!dwpw2_supported(f, g)) continue;
      P2.kind = kKindDwPw2;
      P2.chain_first = p1;
      P1.chained = true;
      P2.bytes_per_frame = (double)D1.spec.cin * D1.ih * D1.iw * 4 + (double)P2.spec.cout * P2.oh * P2.ow * 4;
      P2.flops_per_frame += P1.flops_per_frame;
      P2.weight_bytes += P1.weight_bytes;
    }
  }
  // out = relu(ConvLinear(cat) + shortcut(x)) as ONE 1x1 conv over the channels of both inputs
  // (weights side by side, biases summed): ConvLinear's output, written once and read back as the
  // residual, never exists.  fp32 rounding apart from the two-launch form (one fma chain, not two).
  if (!keep_all && !(flags & UFD_FLAG_NO_RFB_SUM)) {
    Layer& S = m->layers[kRfbShortcut];
    Layer& Lin = m->layers[kRfbLinear];
    if (S.kind == kKindPointwise && Lin.kind == kKindPointwise && S.res_tensor == tensor_of[kRfbLinear] && S.oh == Lin.oh &&
        S.ow == Lin.ow && S.spec.cout == Lin.spec.cout && (Lin.spec.cin & 1) == 0 && (S.spec.cin & 1) == 0) {
      bool only_reader = true;
      for (int j = 0; j < kNumConv; j++)
        if (j != kRfbShortcut && m->layers[j].spec.src == kRfbLinear) only_reader = false;
      if (only_reader) {
        S.sum_with = kRfbLinear;
        S.res_tensor = -1;
        Lin.chained = true;
        S.bytes_per_frame = ((double)Lin.spec.cin + S.spec.cin + S.spec.cout) * S.oh * S.ow * 4;
        S.flops_per_frame += Lin.flops_per_frame;
        S.weight_bytes += Lin.weight_bytes;
      }
    }
  }
  // RFB tail as ONE launch (k_rfb_tail, issued at the shortcut layer's turn): the three dilated 3x3 convs hand their
  // results to the summed 1x1 in registers, the 48-channel concat tensor never exists.  The dilated layers become
  // "chained" (no launch, no output of their own; ufd_debug_layer_output reports them absent in this plan).
  // Handles for a few frames at a time keep the two launches: k_rfb_tail's wave walks three dilated convs and the 56-k-step
  // sum one after the other (26 us for ONE 640x480 frame; the dilated launch + the split-K 1x1 take 8.8 + 10.6).
  m->rfb_tail = false;
  const bool tail_pays = (long)m->B * m->layers[kRfbShortcut].oh * m->layers[kRfbShortcut].ow >= 4L * 60 * 80;
  if (!keep_all && tail_pays && !(flags & UFD_FLAG_NO_RFB_TAIL) && m->layers[kRfbShortcut].sum_with == kRfbLinear) {
    const int dil_layers[3] = {kRfbCatA, kRfbCatB, kRfbCatC};
    ConvArgs d3[3]{}, fin{};
    bool ok = true;
    for (int b = 0; b < 3; b++) {
      const Layer& D = m->layers[dil_layers[b]];
      ok = ok && D.kind == kKindConv3x3 && !D.chained;
      d3[b].k = D.spec.k, d3[b].stride = D.spec.stride, d3[b].dil = D.spec.dil, d3[b].pad = D.spec.pad;
      d3[b].cin = D.spec.cin, d3[b].cout = D.spec.cout, d3[b].relu = D.spec.relu;
      d3[b].ih = D.ih, d3[b].iw = D.iw, d3[b].oh = D.oh, d3[b].ow = D.ow;
    }
    const Layer& S = m->layers[kRfbShortcut];
    fin.k = 1, fin.cout = S.spec.cout, fin.cin = m->layers[kRfbLinear].spec.cin + S.spec.cin;
    fin.in2_ctotal = S.in_tensor >= 0 ? m->tensors[S.in_tensor].c : 0;
    fin.oh = S.oh, fin.ow = S.ow;
    if (ok && rfb_tail_supported(d3, fin)) {
      m->rfb_tail = true;
      Layer& S2 = m->layers[kRfbShortcut];
      for (int b = 0; b < 3; b++) {
        Layer& D = m->layers[dil_layers[b]];
        D.chained = true;
        S2.flops_per_frame += D.flops_per_frame;
        S2.weight_bytes += D.weight_bytes;
      }
    }
  }
  // The three RFB reduce convs (64 -> 8 each, same input) as ONE 64 -> 24 conv: one cout tile instead
  // of three, the input read once; the consumers read channel slices of the stacked tensor.
  if (!keep_all) {
    static const int kStack[3] = {13, 16, 19};
    Layer& A = m->layers[kStack[0]];
    int cout_sum = 0;
    bool ok = true;
    for (int k = 0; k < 3; k++) {
      const Layer& Bm = m->layers[kStack[k]];
      ok = ok && Bm.kind == kKindPointwise && Bm.in_tensor == A.in_tensor && Bm.spec.cin == A.spec.cin && Bm.oh == A.oh &&
           Bm.ow == A.ow && Bm.res_tensor < 0 && Bm.spec.relu == A.spec.relu && Bm.out_coff == 0 && !Bm.chained && Bm.sum_with < 0;
      for (int h = 0; h < 4; h++) ok = ok && kHeadCls[h] != kStack[k] && kHeadReg[h] != kStack[k];
      cout_sum += Bm.spec.cout;
    }
    if (ok && cout_sum <= 32) {
      Tensor t;
      t.c = cout_sum, t.h = A.oh, t.w = A.ow;
      m->tensors.push_back(t);
      const int stacked = (int)m->tensors.size() - 1;
      int coff = 0;
      for (int k = 0; k < 3; k++) {
        Layer& Bm = m->layers[kStack[k]];
        for (int j = 0; j < kNumConv; j++)
          if (m->layers[j].in_tensor == Bm.out_tensor && m->layers[j].spec.src == kStack[k]) m->layers[j].in_tensor = stacked, m->layers[j].in_coff = coff;
        Bm.tap_tensor = stacked, Bm.tap_coff = coff;
        coff += Bm.spec.cout;
        A.stack[k] = kStack[k];
        if (k > 0) {
          Bm.chained = true;
          A.bytes_per_frame += (double)Bm.spec.cout * Bm.oh * Bm.ow * 4;
          A.flops_per_frame += Bm.flops_per_frame;
          A.weight_bytes += Bm.weight_bytes;
        }
      }
      A.out_tensor = stacked;
    }
  }
  // merged launches: layers with identical shapes whose inputs are ready at the leader's turn
  for (int i = 0; i < kNumConv; i++) m->layers[i].leader = i, m->layers[i].group[0] = i, m->layers[i].group[1] = m->layers[i].group[2] = -1;
  {
    // (leader first: the launch is issued at the leader's turn, so a leader that is not the lowest
    // index -- the RFB's dilated convs wait for the b2 branch -- delays the others to its turn)
    static const int kGroups[][3] = {{13, 16, 19}, {14, 17, 20}, {22, 15, 18}, {26, 28, -1}, {36, 38, -1}, {44, 46, -1}, {50, 51, -1}};
    for (const auto& g : kGroups) {
      const Layer& A = m->layers[g[0]];
      bool ok = !A.chained && A.stack[0] < 0;
      for (int k = 1; k < 3 && g[k] >= 0; k++) {
        ok = ok && !m->layers[g[k]].chained;
        const Layer& Bm = m->layers[g[k]];
        ok = ok && Bm.kind == A.kind && Bm.spec.cin == A.spec.cin && Bm.ih == A.ih && Bm.iw == A.iw && Bm.oh == A.oh &&
             Bm.ow == A.ow && Bm.spec.k == A.spec.k && Bm.spec.stride == A.spec.stride &&
             (Bm.spec.cout + 31) / 32 == (A.spec.cout + 31) / 32 && Bm.res_tensor < 0 && A.res_tensor < 0;
        // dense 3x3 convs of stride 1 may differ in dilation (run-time dilation form of the row kernel)
        const bool dil_free = A.kind == kKindConv3x3 && A.spec.stride == 1 && A.ow % 4 == 0 && Bm.spec.pad == Bm.spec.dil &&
                              A.spec.pad == A.spec.dil && Bm.spec.dil <= 5 && A.spec.dil <= 5 && !keep_all;
        ok = ok && (Bm.spec.dil == A.spec.dil || dil_free);
        // one launch at the leader's turn: every member reads the same, already produced tensor, or
        // a tensor whose producing launch comes before that turn
        const int src_a = A.kind == kKindDwPw ? m->layers[A.fused_dw].in_tensor : A.in_tensor;
        const int src_b = Bm.kind == kKindDwPw ? m->layers[Bm.fused_dw].in_tensor : Bm.in_tensor;
        const int prod_b = Bm.spec.src;
        const bool produced_before = Bm.kind == kKindConv3x3 && prod_b >= 0 && prod_b < g[0] && m->layers[prod_b].leader < g[0] &&
                                     g[k] < g[0];
        ok = ok && (src_a == src_b || produced_before);
        if (A.kind == kKindDwPw)
          ok = ok && m->layers[Bm.fused_dw].spec.stride == m->layers[A.fused_dw].spec.stride &&
               m->layers[Bm.fused_dw].ih == m->layers[A.fused_dw].ih;
        if (A.kind == kKindConv3x3) ok = ok && Bm.spec.cout <= 16 && A.spec.cout <= 16;  // (row kernel or gather kernel: one cout tile)
      }
      if (!ok) continue;
      for (int k = 0; k < 3 && g[k] >= 0; k++) {
        m->layers[g[k]].leader = g[0];
        m->layers[g[0]].group[k] = g[k];
      }
    }
  }
  // Dual launches: a cls/reg head pair and the next backbone block both read the tensor produced just before them and
  // do not depend on each other -- one grid for both (k_dual_*), issued at the head pair's turn.
  if (!keep_all && !(flags & UFD_FLAG_NO_DUAL)) {
    static const int kDuals[][2] = {{kHeadCls[0], 30}, {kHeadCls[1], 40}, {kHeadCls[2], 47}};
    for (const auto& d : kDuals) {
      Layer& A = m->layers[d[0]];
      Layer& Bm = m->layers[d[1]];
      if (A.kind != kKindDwPw || A.leader != d[0] || A.chained) continue;
      if ((Bm.kind != kKindDwPw && Bm.kind != kKindPointwise) || Bm.leader != d[1] || Bm.group[1] >= 0 || Bm.chained ||
          Bm.stack[0] >= 0 || Bm.sum_with >= 0 || Bm.res_tensor >= 0 || d[1] <= d[0])
        continue;
      // everything B reads exists before A's turn
      const int src = Bm.kind == kKindDwPw ? m->layers[Bm.fused_dw].spec.src : Bm.spec.src;
      if (src < 0 || src >= d[0] || m->layers[src].chained) continue;
      const int src_turn = m->layers[src].ride >= 0 ? m->layers[src].ride : m->layers[src].leader;
      if (src_turn >= d[0]) continue;
      A.rider = d[1];
      Bm.ride = d[0];
    }
  }
  // where each layer's output can be read back in this plan (ufd_debug_layer_output)
  for (int i = 0; i < kNumConv; i++) {
    Layer& L = m->layers[i];
    if (L.tap_tensor >= 0) continue;  // slice of the stacked reduce tensor
    if ((L.kind == kKindFusedAway && !L.materialize) || L.chained) continue;
    L.tap_tensor = L.out_tensor, L.tap_coff = L.out_coff;
  }
  // liveness: first writer, last reader (head outputs live until the decode kernel)
  const int nt = (int)m->tensors.size();
  std::vector<int> first(nt, kNumConv), last(nt, -1);
  for (int i = 0; i < kNumConv; i++) {
    const Layer& L = m->layers[i];
    if (L.kind == kKindFusedAway && !L.materialize) continue;  // never written, never read
    if (L.chained) continue;                                     // computed inside a later launch
    const int lead = m->layers[L.leader].ride >= 0 ? std::min(L.leader, m->layers[L.leader].ride) : L.leader;
    first[L.out_tensor] = std::min(first[L.out_tensor], lead);  // a merged layer writes at its leader's turn, a rider at its host's
    last[L.out_tensor] = std::max(last[L.out_tensor], std::max(i, L.leader));
    const int when = std::max(i, L.leader);  // a merged member is read at its leader's turn
    int src_t = L.kind == kKindDwPw ? m->layers[L.fused_dw].in_tensor : L.in_tensor;
    if (L.kind == kKindDwPw2) src_t = m->layers[m->layers[L.chain_first].fused_dw].in_tensor;
    if (src_t >= 0) last[src_t] = std::max(last[src_t], when);
    if (L.in_tensor >= 0 && L.kind != kKindDwPw && L.kind != kKindDwPw2) last[L.in_tensor] = std::max(last[L.in_tensor], when);
    if (L.res_tensor >= 0) last[L.res_tensor] = std::max(last[L.res_tensor], when);
    if (L.sum_with >= 0 && m->layers[L.sum_with].in_tensor >= 0)
      last[m->layers[L.sum_with].in_tensor] = std::max(last[m->layers[L.sum_with].in_tensor], when);
  }
  if (m->rfb_tail)  // the fused launch reads the dilated convs' inputs at the shortcut layer's turn
    for (int j : {kRfbCatA, kRfbCatB, kRfbCatC})
      if (m->layers[j].in_tensor >= 0) last[m->layers[j].in_tensor] = std::max(last[m->layers[j].in_tensor], kRfbShortcut);
  for (int h = 0; h < 4; h++) {
    last[tensor_of[kHeadCls[h]]] = kNumConv;
    last[tensor_of[kHeadReg[h]]] = kNumConv;
  }
  struct Blk {
    size_t off, size;
  };
  std::vector<Blk> free_list;
  size_t top = 0;
  auto align = [](size_t v) { return (v + 63) & ~(size_t)63; };
  auto allocate = [&](int t) {
    const size_t need = align(m->tensors[t].per_frame() * m->B);
    size_t best = (size_t)-1;
    for (size_t i = 0; i < free_list.size(); i++)
      if (free_list[i].size >= need && (best == (size_t)-1 || free_list[i].size < free_list[best].size)) best = i;
    if (best != (size_t)-1) {
      m->tensors[t].off = free_list[best].off;
      if (free_list[best].size > need) {
        free_list[best].off += need;
        free_list[best].size -= need;
      } else {
        free_list.erase(free_list.begin() + best);
      }
    } else {
      m->tensors[t].off = top;
      top += need;
    }
  };
  std::vector<bool> allocated(nt, false);
  for (int i = 0; i < kNumConv; i++) {
    for (int j = i; j < kNumConv; j++) {  // every tensor first written at turn i (merged members included)
      const int tj = m->layers[j].out_tensor;
      if (first[tj] == i && !allocated[tj]) {
        allocate(tj);
        allocated[tj] = true;
      }
    }  // (a fused-away depthwise output has first == kNumConv: no storage)
    // a buffer is recycled only after the layer that reads it last has been issued, so a
    // layer's output never aliases its own inputs
    if (!keep_all)
      for (int u = 0; u < nt; u++)
        if (last[u] == i && allocated[u])  // (a tensor no launch writes -- the RFB concat under k_rfb_tail -- has no storage to give back)
          free_list.push_back({m->tensors[u].off, align(m->tensors[u].per_frame() * m->B)});
  }
  for (int u = 0; u < nt; u++) m->tensors[u].first = first[u], m->tensors[u].last = last[u], m->tensors[u].stored = allocated[u];
  m->arena_floats = top;
}

int upload_weights(ufd_model* m, const float* blob) {
  const ConvSpec* specs = conv_specs();
  std::vector<float> img;
  std::vector<size_t> w_off(kNumConv), b_off(kNumConv), dw_off(kNumConv, (size_t)-1), rows_off(kNumConv, (size_t)-1);
  const float* p = blob;
  for (int i = 0; i < kNumConv; i++) {
    const ConvSpec& s = specs[i];
    const size_t nw = conv_weight_floats(s);
    while (img.size() % 64) img.push_back(0.f);
    w_off[i] = img.size();
    const LayerKind kind = m->layers[i].kind;
    if (kind == kKindPointwise || kind == kKindDwPw || kind == kKindDwPw2) {
      const size_t np = pointwise_packed_floats(s.cin, s.cout);
      img.resize(img.size() + np);
      pack_pointwise_weights(p, s.cin, s.cout, img.data() + w_off[i]);
    } else if (kind == kKindConv3x3) {
      const size_t np = conv3x3_packed_floats(s.cin);
      img.resize(img.size() + np);
      pack_conv3x3_weights(p, s.cin, s.cout, img.data() + w_off[i]);
    } else {
      img.insert(img.end(), p, p + nw);
    }
    p += nw;
    while (img.size() % 64) img.push_back(0.f);
    b_off[i] = img.size();
    img.insert(img.end(), p, p + s.cout);
    if (kind == kKindConv3x3) {
      while (img.size() % 64) img.push_back(0.f);
      rows_off[i] = img.size();
      img.resize(img.size() + conv3x3_rows_packed_floats(s.cin));
      pack_conv3x3_rows_weights(p - nw, s.cin, s.cout, img.data() + rows_off[i]);
    }
    if (s.groups > 1 && s.k == 3) {  // depthwise: also the [c][12] image the fused kernel copies into LDS
      while (img.size() % 64) img.push_back(0.f);
      dw_off[i] = img.size();
      img.resize(img.size() + depthwise_packed_floats(s.cout));
      pack_depthwise_weights(p - nw, p, s.cout, img.data() + dw_off[i]);
    }
    p += s.cout;
  }
  // summed 1x1 pairs: weights of both convs side by side per output channel, biases added
  std::vector<size_t> sumw_off(kNumConv, (size_t)-1), sumb_off(kNumConv, (size_t)-1);
  {
    std::vector<const float*> wsrc(kNumConv), bsrc(kNumConv);
    const float* q = blob;
    for (int i = 0; i < kNumConv; i++) {
      wsrc[i] = q;
      q += conv_weight_floats(specs[i]);
      bsrc[i] = q;
      q += specs[i].cout;
    }
    for (int i = 0; i < kNumConv; i++) {
      if (m->layers[i].stack[0] != i) continue;
      const int ci = specs[i].cin;
      std::vector<float> wcat, bcat;
      for (int k : m->layers[i].stack) {
        if (k < 0) continue;
        wcat.insert(wcat.end(), wsrc[k], wsrc[k] + (size_t)specs[k].cout * ci);
        bcat.insert(bcat.end(), bsrc[k], bsrc[k] + specs[k].cout);
      }
      while (img.size() % 64) img.push_back(0.f);
      sumw_off[i] = img.size();
      img.resize(img.size() + pointwise_packed_floats(ci, (int)bcat.size()));
      pack_pointwise_weights(wcat.data(), ci, (int)bcat.size(), img.data() + sumw_off[i]);
      while (img.size() % 64) img.push_back(0.f);
      sumb_off[i] = img.size();
      img.insert(img.end(), bcat.begin(), bcat.end());
    }
    for (int i = 0; i < kNumConv; i++) {
      const int j = m->layers[i].sum_with;
      if (j < 0) continue;
      const int ca = specs[j].cin, cb = specs[i].cin, co = specs[i].cout;
      std::vector<float> wcat((size_t)co * (ca + cb)), bsum(co);
      for (int o = 0; o < co; o++) {
        std::memcpy(&wcat[(size_t)o * (ca + cb)], wsrc[j] + (size_t)o * ca, sizeof(float) * ca);
        std::memcpy(&wcat[(size_t)o * (ca + cb) + ca], wsrc[i] + (size_t)o * cb, sizeof(float) * cb);
        bsum[o] = bsrc[j][o] + bsrc[i][o];
      }
      while (img.size() % 64) img.push_back(0.f);
      sumw_off[i] = img.size();
      img.resize(img.size() + pointwise_packed_floats(ca + cb, co));
      pack_pointwise_weights(wcat.data(), ca + cb, co, img.data() + sumw_off[i]);
      while (img.size() % 64) img.push_back(0.f);
      sumb_off[i] = img.size();
      img.insert(img.end(), bsum.begin(), bsum.end());
    }
  }
  size_t tail_off = (size_t)-1;
  if (m->rfb_tail) {
    const float* q = blob;
    const float *w_lin = nullptr, *w_short = nullptr;
    for (int i = 0; i < kNumConv; i++) {
      if (i == kRfbLinear) w_lin = q;
      if (i == kRfbShortcut) w_short = q;
      q += conv_weight_floats(specs[i]) + specs[i].cout;
    }
    while (img.size() % 64) img.push_back(0.f);
    tail_off = img.size();
    img.resize(img.size() + rfb_tail_packed_floats());
    pack_rfb_tail_weights(w_lin, w_short, img.data() + tail_off);
  }
  m->weight_img_floats = img.size();
  HIPC(m, hipMalloc(&m->d_weights, img.size() * sizeof(float)));
  HIPC(m, hipMemcpy(m->d_weights, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
  for (int i = 0; i < kNumConv; i++) {
    m->layers[i].d_w = m->d_weights + w_off[i];
    m->layers[i].d_b = m->d_weights + b_off[i];
    if (dw_off[i] != (size_t)-1) m->layers[i].d_w_dwpack = m->d_weights + dw_off[i];
    if (rows_off[i] != (size_t)-1) m->layers[i].d_w_rows = m->d_weights + rows_off[i];
    if (sumw_off[i] != (size_t)-1) m->layers[i].d_w_sum = m->d_weights + sumw_off[i], m->layers[i].d_b_sum = m->d_weights + sumb_off[i];
  }
  if (tail_off != (size_t)-1) m->layers[kRfbShortcut].d_w_tail = m->d_weights + tail_off;
  return UFD_OK;
}

float* tensor_ptr(ufd_model* m, int t) { return tl_cur->d_arena + m->tensors[t].off; }

// one conv layer for frames [f0, f0 + count) of the batch
// Kernel arguments of conv layer i for frames [f0, f0 + count)
ConvArgs layer_args(ufd_model* m, int i, uint32_t f0, uint32_t count, int* dw_stride) {
  const Layer& L = m->layers[i];
  auto in_ptr = [&](int t, int ih, int iw) -> const float* {
    if (t < 0) return tl_cur->d_input + (size_t)f0 * 3 * ih * iw;
    return tensor_ptr(m, t) + (size_t)f0 * m->tensors[t].per_frame();
  };
  ConvArgs a{};
  a.in = in_ptr(L.in_tensor, L.ih, L.iw);
  a.w = L.d_w;
  a.bias = L.d_b;
  a.out = L.chained ? nullptr : tensor_ptr(m, L.out_tensor) + (size_t)f0 * m->tensors[L.out_tensor].per_frame();
  a.res = L.res_tensor >= 0 ? in_ptr(L.res_tensor, 0, 0) : nullptr;
  a.B = (int)count;
  a.cin = L.spec.cin, a.cout = L.spec.cout;
  a.ih = L.ih, a.iw = L.iw, a.oh = L.oh, a.ow = L.ow;
  a.k = L.spec.k, a.stride = L.spec.stride, a.pad = L.spec.pad, a.dil = L.spec.dil;
  a.depthwise = L.spec.groups > 1;
  a.relu = L.spec.relu || i == kRfbShortcut;
  a.in_ctotal = L.in_tensor < 0 ? 3 : m->tensors[L.in_tensor].c;
  a.out_ctotal = m->tensors[L.out_tensor].c;
  a.out_coff = L.out_coff;
  *dw_stride = 1;
  if (L.in_coff) a.in += (size_t)L.in_coff * L.ih * L.iw;  // a channel slice of a stacked tensor
  if (L.stack[0] == i) {  // this launch computes the stacked output channels of all members
    a.cout = m->tensors[L.out_tensor].c;
    a.w = L.d_w_sum;
    a.bias = L.d_b_sum;
  }
  if (L.sum_with >= 0) {  // two summed 1x1 convs: first the other conv's input channels, then this layer's
    const Layer& O = m->layers[L.sum_with];
    a.in2 = a.in;
    a.in2_ctotal = a.in_ctotal;
    a.in = in_ptr(O.in_tensor, O.ih, O.iw);
    a.in_ctotal = O.in_tensor < 0 ? 3 : m->tensors[O.in_tensor].c;
    a.ksplit = O.spec.cin >> 1;
    a.cin = O.spec.cin + L.spec.cin;
    a.w = L.d_w_sum;
    a.bias = L.d_b_sum;
  }
  if (L.kind == kKindDwPw2) {  // second block of a chain: its input tensor does not exist
    const Layer& D = m->layers[L.fused_dw];
    a.in = nullptr;
    a.ih = D.ih, a.iw = D.iw;
    a.w2 = D.d_w_dwpack, a.bias2 = D.d_b;
    *dw_stride = D.spec.stride;
  }
  if (L.kind == kKindDwPw) {
    const Layer& D = m->layers[L.fused_dw];
    a.in = in_ptr(D.in_tensor, D.ih, D.iw);
    a.in_ctotal = D.in_tensor < 0 ? 3 : m->tensors[D.in_tensor].c;
    a.ih = D.ih, a.iw = D.iw;
    a.w2 = D.d_w_dwpack, a.bias2 = D.d_b;
    *dw_stride = D.spec.stride;
  }
  return a;
}

// Issues conv layer i -- together with the layers merged into its launch (Layer::group: cls + reg
// head pairs and the three RFB reduce convs share shapes and run as one launch, blockIdx.y
// selecting the member).  Non-leading members are skipped when their turn comes.
void tap_outputs(ufd_model* m, int i, uint32_t count, hipStream_t st);

void enqueue_layer_launch(ufd_model* m, int i, uint32_t f0, uint32_t count, hipStream_t st);

void enqueue_layer(ufd_model* m, int i, uint32_t count) {
  hipStream_t st = tl_cur->stream;
  enqueue_layer_launch(m, i, 0, count, st);
  if (!m->tap_buf.empty() && tl_cur == &m->ctx[0]) tap_outputs(m, i, count, st);
}

// UFD_FLAG_TAP_LAYERS: copies every tensor the launch issued at layer i's turn has just written
// (whole batch) to its tap buffer, before the arena recycles it.
void tap_outputs(ufd_model* m, int i, uint32_t count, hipStream_t st) {
  const Layer& L = m->layers[i];
  if ((L.kind == kKindFusedAway && !L.materialize) || L.chained || L.leader != i) return;
  int seen[3] = {-1, -1, -1}, n = 0;
  for (int j : L.group) {
    if (j < 0) continue;
    const int t = m->layers[j].out_tensor;
    if (t == seen[0] || t == seen[1]) continue;
    seen[n++] = t;
    (void)hipMemcpyAsync(m->tap_buf[t], tensor_ptr(m, t), sizeof(float) * m->tensors[t].per_frame() * count, hipMemcpyDeviceToDevice, st);
  }
}

// Timing experiments only (UFD_ABLATE_LAYERS=4,8: skip the launches issued at those layer indices; the tensors they would
// have written keep whatever the arena held, so results are garbage): what a launch costs the loaded pipeline is the frame
// rate without it -- the upper bound of what making it faster can give (docs/EXPERIMENTS.md, round 5).
static bool ablated(int i) {
  static const std::vector<int> skip = [] {
    std::vector<int> v;
    if (const char* e = experiment_env("UFD_ABLATE_LAYERS"))
      for (const char* p = e; *p;) {
        v.push_back(std::atoi(p));
        while (*p && *p != ',') p++;
        if (*p) p++;
      }
    return v;
  }();
  return std::find(skip.begin(), skip.end(), i) != skip.end();
}

void enqueue_layer_launch(ufd_model* m, int i, uint32_t f0, uint32_t count, hipStream_t st) {
  const Layer& L = m->layers[i];
  if (ablated(i)) return;
  if (L.kind == kKindFusedAway && !L.materialize) return;
  if (L.chained) return;      // computed inside the kKindDwPw2 launch of the next block
  if (L.leader != i) return;  // issued with its group leader
  if (L.ride >= 0 && !tl_force_rider) return;  // issued in (or right behind) the launch of the layer it rides with
  if (i == 0 && tl_cur->stem_descs) {  // stem conv straight from the decoder's sample planes
    int st_ = 1;
    StemArgs sa;
    sa.a = layer_args(m, 0, f0, count, &st_);
    sa.a.w = L.d_w_rows;
    sa.descs = tl_cur->stem_descs + f0;
    sa.planes = tl_cur->d_planes + (size_t)f0 * m->plane_stride;
    sa.plane_stride = m->plane_stride;
    sa.lut = m->d_lut;
    ProfScope ps(m, std::string("stem_planes_mfma:") + L.spec.name,
                 (double)count * (1.5 * L.ih * L.iw + 4.0 * L.spec.cout * L.oh * L.ow) + L.weight_bytes, L.flops_per_frame * count, st);
    launch_stem_planes_mfma(sa, st);
    return;
  }
  if (i == kRfbShortcut && m->rfb_tail) {  // the three dilated convs + relu(ConvLinear(cat) + shortcut(x)) as one launch
    int st_ = 1;
    ConvArgs d3[3];
    const int dil_layers[3] = {kRfbCatA, kRfbCatB, kRfbCatC};
    std::string names;
    for (int b = 0; b < 3; b++) {
      d3[b] = layer_args(m, dil_layers[b], f0, count, &st_);
      d3[b].w = m->layers[dil_layers[b]].d_w_rows;
      names += std::string(m->layers[dil_layers[b]].spec.name) + "+";
    }
    ConvArgs fin = layer_args(m, i, f0, count, &st_);
    fin.w = L.d_w_tail;
    ProfScope ps(m, std::string("rfb_tail:") + names + L.spec.name,
                 L.bytes_per_frame * count + L.weight_bytes, L.flops_per_frame * count, st);
    launch_rfb_tail(d3, fin, st);
    return;
  }
  if (L.kind == kKindDwPw2) {
    int s1 = 1, s2 = 2;
    const Layer& F = m->layers[L.chain_first];
    const ConvArgs first = layer_args(m, L.chain_first, f0, count, &s1);
    const ConvArgs second = layer_args(m, i, f0, count, &s2);
    ProfScope ps(m, std::string("conv_dwpw2_mfma") + conv_dwpw2_instance(first, second) + ":" + F.spec.name + "+" + L.spec.name,
                 L.bytes_per_frame * count + L.weight_bytes,
                 L.flops_per_frame * count, st);
    launch_conv_dwpw2_mfma(first, second, st);
    return;
  }
  ConvArgs args[3];
  int n = 0, dw_stride = 1;
  std::string names;
  double bytes = 0, flops = 0;
  for (int j : L.group) {
    if (j < 0) continue;
    const Layer& M = m->layers[j];
    args[n++] = layer_args(m, j, f0, count, &dw_stride);
    names += (names.empty() ? "" : "+") + std::string(M.spec.name);
    bytes += M.bytes_per_frame * count + M.weight_bytes;
    flops += M.flops_per_frame * count;
  }
  ConvArgs& a = args[0];
  bool use_rows = false;
  const char* kind = "conv_direct_full";
  switch (L.kind) {
    case kKindPointwise: kind = "conv_pw_mfma"; break;
    case kKindDwPw: kind = dwpw_uses_coop(args, n) ? "conv_dwpw_coop" : "conv_dwpw_mfma"; break;
    case kKindDwPw2: break;  // issued above
    case kKindConv3x3:
      use_rows = true;
      for (int j = 0; j < n; j++) use_rows = use_rows && conv3x3_rows_supported(args[j]);
      kind = use_rows ? "conv3x3_rows_mfma" : "conv3x3_mfma";
      break;
    case kKindFusedAway: kind = "conv_direct_dw_debug"; break;
    case kKindDirect: kind = a.depthwise ? "conv_direct_dw" : "conv_direct_full"; break;
  }
  if (L.rider >= 0 && L.kind == kKindDwPw) {  // dual launch with the rider's conv, when that pair of instances exists
    const Layer& R = m->layers[L.rider];
    int r_stride = 1;
    const ConvArgs rb = layer_args(m, L.rider, f0, count, &r_stride);
    const int rb_stride = R.kind == kKindDwPw ? r_stride : 0;
    // (labelled with the device function and its template instance, like every other launch: "conv_dual_coop<1, 1, 2>")
    if (const char* label = conv_dual_instance(args, n, dw_stride, &rb, rb_stride)) {
      ProfScope ps(m, std::string(label) + ":" + names + "|" + R.spec.name, bytes + R.bytes_per_frame * count + R.weight_bytes,
                   flops + R.flops_per_frame * count, st);
      if (launch_conv_dual(args, n, dw_stride, &rb, rb_stride, st)) return;
      ps.cancel();
    }
  }
  const char* inst = L.kind == kKindPointwise ? conv_pointwise_instance(args, n)
                     : (L.kind == kKindDwPw ? conv_dwpw_instance(args, n, dw_stride) : (use_rows ? conv3x3_rows_instance(args, n) : ""));
  {
  ProfScope ps(m, std::string(kind) + inst + ":" + names, bytes, flops, st);
  switch (L.kind) {
    case kKindPointwise: launch_conv_pointwise_mfma(args, n, st); break;
    case kKindDwPw: launch_conv_dwpw_mfma(args, n, dw_stride, st); break;
    case kKindConv3x3:
      if (use_rows) {
        int k = 0;
        for (int j : L.group)
          if (j >= 0) args[k++].w = m->layers[j].d_w_rows;
        launch_conv3x3_rows_mfma(args, n, st);
      } else {
        launch_conv3x3_mfma(args, n, st);
      }
      break;
    default: launch_conv_direct(a, st); break;
  }
  }
  if (L.rider >= 0) {  // the pair is not compiled as one grid: the rider right behind its host, on its own
    tl_force_rider = true;
    enqueue_layer_launch(m, L.rider, f0, count, st);
    tl_force_rider = false;
  }
}

}  // namespace ufd

extern "C" {

int ufd_debug_plan(uint32_t variant, uint32_t max_batch, uint32_t flags, ufd_plan_layer* layers, uint32_t layer_cap, uint32_t* n_layers,
                   ufd_plan_tensor* tensors, uint32_t tensor_cap, uint32_t* n_tensors, uint64_t* arena_floats) {
  if ((variant != 640 && variant != 320) || !max_batch || !n_layers || !n_tensors) return UFD_E_ARG;
  try {
    std::unique_ptr<ufd_model> m(new ufd_model());
    m->cfg.variant = variant, m->cfg.flags = flags, m->cfg.max_batch = max_batch;
    m->W = variant == 640 ? 640 : 320, m->H = variant == 640 ? 480 : 240;
    m->B = max_batch;
    plan_tensors(m.get(), (flags & UFD_FLAG_KEEP_LAYERS) != 0);
    *n_layers = (uint32_t)m->layers.size(), *n_tensors = (uint32_t)m->tensors.size();
    if (arena_floats) *arena_floats = m->arena_floats;
    for (uint32_t i = 0; i < *n_layers && i < layer_cap && layers; i++) {
      const Layer& L = m->layers[i];
      ufd_plan_layer& o = layers[i];
      std::memset(&o, 0, sizeof(o));
      std::snprintf(o.name, sizeof(o.name), "%s", L.spec.name);
      o.kind = (int32_t)L.kind, o.leader = L.leader, o.ride = L.ride, o.chain_first = L.chain_first, o.fused_dw = L.fused_dw;
      o.chained = L.chained ? 1 : 0, o.materialize = L.materialize ? 1 : 0;
      o.in_tensor = L.in_tensor, o.out_tensor = L.out_tensor, o.out_coff = L.out_coff, o.tap_tensor = L.tap_tensor;
      // issues a launch of its own at its turn: not computed inside another launch, not a non-leading member, not a rider
      o.launches = !(L.kind == kKindFusedAway && !L.materialize) && !L.chained && L.leader == (int)i && L.ride < 0;
      if ((int)i == kRfbShortcut && m->rfb_tail) o.rfb_tail = 1;
    }
    for (uint32_t t = 0; t < *n_tensors && t < tensor_cap && tensors; t++) {
      const Tensor& T = m->tensors[t];
      tensors[t] = ufd_plan_tensor{(uint64_t)T.off, (uint64_t)T.per_frame() * max_batch, T.c, T.h, T.w, T.first, T.last, T.stored ? 1 : 0};
    }
    return UFD_OK;
  } catch (...) {
    return UFD_E_DEVICE;
  }
}

}  // extern "C"
