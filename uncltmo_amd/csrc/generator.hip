// Whole-generator forward: enqueues every kernel of the published UNet (depth 4, 32 filters, transposed-conv
// decoder, "square_and_square_root" skip operator, ViG bottleneck) on one stream.
// Follows Unet_singleFrame.UNet.forward (Unet_singleFrame.py:177-213) / Unet.UNet.forward (Unet.py:213-289).
//
// Tiles are processed in chunks sized so that a layer's output is still resident in the 256 MiB Infinity
// Cache when the next layer reads it; skip tensors and (optionally) all activations stay in the workspace.
#include <map>
#include <mutex>

#include <cstdlib>

#include "common.h"
#include "bwd_internal.h"

// 2 (default): fc1 and the kNN graph as their own launches, the rest of the block fused (uncl_gcn_tail); 1: the whole block as
// one launch (uncl_gcn_block: measured no faster -- 200 us against 36 + 85 + 84 -- the phases of one sample serialise inside
// its workgroup either way); 0: separate kernels
static int g_fused_graph = [] { const char* e = getenv("UNCL_GCN_FUSED"); return e ? atoi(e) : 2; }();

// bf16 backward pass without float atomics (deterministic); env UNCL_BWD_DET sets the initial state (default 0: atomics, faster)
static std::atomic<int> g_bwd_det{[] { const char* e = getenv("UNCL_BWD_DET"); return e ? atoi(e) : 0; }()};
extern "C" int uncl_gen_set_deterministic(int on) { return g_bwd_det.exchange(on ? 1 : 0); }
// 1: inference runs the last decoder stage (up_path.3: concat + fused up-conv -> ConvT3x3 -> ConvT3x3 -> outconv) as ONE launch whose
// 32-channel maps stay in LDS (conv3x3_pc.hip, TAIL): 1.9 GB less HBM traffic per 200 tiles; 0 (default): two launches with the
// 254 x 254 x 32 map in HBM between them.  Default by measured time: the fused launch takes 1.17 ms against 0.72 + 0.32, the whole
// step ties (same-box A/B -0.8 % on one box, +0.4 % on another)
#ifndef UNCL_HYBRID_DEFAULT
#define UNCL_HYBRID_DEFAULT 0
#endif
#ifndef UNCL_SPLIT_PCT_DEFAULT
#define UNCL_SPLIT_PCT_DEFAULT 50
#endif
static int g_fuse_tail = [] { const char* e = getenv("UNCL_FUSE_TAIL"); return e ? atoi(e) : 0; }();
// inference: up_path.2.up (64 -> 64 channels) recomputed inside up_path.2.conv.conv's loader (conv3x3_pc.hip, MODE 5)
static int g_fuse_up64 = [] { const char* e = getenv("UNCL_FUSE_UP64"); return e ? atoi(e) : 1; }();
extern "C" int uncl_gen_set_fused_tail(int on) {
  const int old = g_fuse_tail;
  g_fuse_tail = on ? 1 : 0;
  return old;
}

namespace {

// spatial size of every stage for a 256x256 input
constexpr int S_IN = 256;
constexpr int S_INC0 = 254, S_X0 = 252;   // inc
constexpr int S_D0A = 124, S_X1 = 122;    // down0 (pool 126)
constexpr int S_D1A = 59, S_X2 = 57;      // down1 (pool 61)
constexpr int S_D2A = 26, S_X3 = 24;      // down2 (pool 28)
constexpr int S_D3A = 10, S_X4 = 12;      // down3 (pool 12): conv -> 10, convT -> 12
constexpr int NODES = 144;

enum Buf {
  B_INC0, B_X0, B_D0A, B_X1, B_D1A, B_X2, B_D2A, B_X3, B_D3A, B_X4,
  B_GFC1, B_GMR, B_GGC, B_GX1, B_FH, B_GOUT,
  B_U0UP, B_U0A, B_U0, B_U1UP, B_U1A, B_U1, B_U2UP, B_U2A, B_U2, B_U3UP, B_U3A, B_UPX,
  B_KNN, B_X0P, B_X1P, B_X2P, B_X3P, B_GGCZ, B_FHZ,
  // InstanceNorm training mode only: the normalised pre-activation of every 3x3 conv output (same extent as that output) and
  // 1/std per (sample, channel) of the 18 conv layers, fp32
  B_ZN0, B_ZN_END = B_ZN0 + 18, B_RSTD = B_ZN_END, B_COUNT
};
// the 18 convolutions that a norm follows, as the activation buffer each writes
constexpr int kNormBuf[18] = {B_INC0, B_X0, B_D0A, B_X1, B_D1A, B_X2, B_D2A, B_X3, B_D3A, B_X4,
                              B_U0A, B_U0, B_U1A, B_U1, B_U2A, B_U2, B_U3A, B_UPX};
constexpr int RSTD_FLOATS = 32 + 32 + 64 + 64 + 128 + 128 + 256 + 256 + 256 + 256 + 128 + 128 + 64 + 64 + 32 + 32 + 32 + 32;
// BatchNorm training: behind the 1/std blocks, per sample, room for the fp64 partial sums of one layer ([256 channels][2]) and a
// share of the per-channel coefficients (2 x 256 floats per CALL): bwd_bnorm_forward / _backward scratch
constexpr int BN_SCRATCH_FLOATS = 1024 + 512;
inline int norm_index(int buf) {
  for (int i = 0; i < 18; ++i)
    if (kNormBuf[i] == buf) return i;
  return -1;
}

struct BufDim { int h, w, c; };
const BufDim kDims[B_COUNT] = {
    {254, 254, 32}, {252, 252, 32}, {124, 124, 64}, {122, 122, 64}, {59, 59, 128}, {57, 57, 128},
    {26, 26, 256},  {24, 24, 256},  {10, 10, 256},  {12, 12, 256},
    {1, 144, 256},  {1, 144, 512},  {1, 144, 512},  {1, 144, 256},  {1, 144, 256}, {1, 144, 256},
    {24, 24, 256},  {26, 26, 128},  {28, 28, 128},  {56, 56, 128},  {59, 59, 64},  {61, 61, 64},
    {122, 122, 64}, {124, 124, 32}, {126, 126, 32}, {252, 252, 32}, {254, 254, 32}, {256, 256, 32},
    {1, 144, 9},
    {126, 126, 32}, {61, 61, 64}, {28, 28, 128}, {12, 12, 256},
    {1, 144, 512}, {1, 144, 256},
    // B_ZN0 ..: dims of kNormBuf[i]
    {254, 254, 32}, {252, 252, 32}, {124, 124, 64}, {122, 122, 64}, {59, 59, 128}, {57, 57, 128}, {26, 26, 256}, {24, 24, 256},
    {10, 10, 256}, {12, 12, 256}, {26, 26, 128}, {28, 28, 128}, {59, 59, 64}, {61, 61, 64}, {124, 124, 32}, {126, 126, 32},
    {254, 254, 32}, {256, 256, 32},
    {1, RSTD_FLOATS + BN_SCRATCH_FLOATS, 1}};

struct Layout {
  size_t off[B_COUNT];
  size_t per_n[B_COUNT];  // bytes per tile
  size_t total;
};

Layout make_layout(int n_alloc, int dtype, int norm_train = 0) {
  const size_t es = uncl_is_h16(dtype) ? 2 : 4;
  Layout L;
  size_t o = 0;
  for (int b = 0; b < B_COUNT; ++b) {
    const size_t e = (b == B_KNN || b == B_RSTD) ? 4 : es;
    L.per_n[b] = (size_t)kDims[b].h * kDims[b].w * kDims[b].c * e;
    if (b >= B_ZN0 && b < B_ZN_END && !norm_train) L.per_n[b] = 0;
    L.off[b] = o;
    o += (L.per_n[b] * n_alloc + 255) & ~(size_t)255;
  }
  L.total = o;
  return L;
}

const char* kNames[] = {
    "inc.conv.conv1",
    "down_path.0.mpconv.1.conv", "down_path.0.mpconv.1.conv1",
    "down_path.1.mpconv.1.conv", "down_path.1.mpconv.1.conv1",
    "down_path.2.mpconv.1.conv", "down_path.2.mpconv.1.conv1",
    "down_path.3.mpconv.1.conv", "down_path.3.mpconv.1.conv1",
    "gcn.module.0.0.fc1.0", "gcn.module.0.0.graph_conv.gconv.nn.0", "gcn.module.0.0.fc2.0",
    "gcn.module.0.1.fc1.0", "gcn.module.0.1.fc2.0",
    "up_path.0.up", "up_path.0.conv.conv", "up_path.0.conv.conv1",
    "up_path.1.up", "up_path.1.conv.conv", "up_path.1.conv.conv1",
    "up_path.2.up", "up_path.2.conv.conv", "up_path.2.conv.conv1",
    "up_path.3.up", "up_path.3.conv.conv", "up_path.3.conv.conv1"};
static_assert(sizeof(kNames) / sizeof(kNames[0]) == UNCL_G_NUM_WEIGHTS, "weight table size");

enum W {
  W_INC1, W_D0A, W_D0B, W_D1A, W_D1B, W_D2A, W_D2B, W_D3A, W_D3B,
  W_GFC1, W_GGC, W_GFC2, W_FFC1, W_FFC2,
  W_U0UP, W_U0A, W_U0B, W_U1UP, W_U1A, W_U1B, W_U2UP, W_U2A, W_U2B, W_U3UP, W_U3A, W_U3B
};

// Optional per-layer timing (bench.py's roofline leg): HIP events around every launch of ONE packed-weight
// layer, on the stream the kernels run on.  Off unless uncl_prof_enable() was called.
struct Prof {
  int layer = -1;
  int cap = 0, count = 0;
  hipEvent_t* ev = nullptr;  // 2*cap events
} g_prof;

struct ProfScope {
  bool on;
  hipStream_t s;
  ProfScope(int layer, hipStream_t st) : on(layer == g_prof.layer && g_prof.count < g_prof.cap), s(st) {
    if (on) (void)hipEventRecord(g_prof.ev[2 * g_prof.count], s);
  }
  ~ProfScope() {
    if (on) { (void)hipEventRecord(g_prof.ev[2 * g_prof.count + 1], s); ++g_prof.count; }
  }
};

// unet_norm = 'batch_norm', training mode: the eighteen BatchNorm2d layers' parameters / running statistics / gradient slots, in the
// order of kNormBuf.  Thread-local, set by uncl_gen_set_bn before the forward / backward calls that use it (w->norm == 2).
struct BnState {
  const float* gamma[18];
  const float* beta[18];
  float* rmean[18];
  float* rvar[18];
  float* g_gamma[18];
  float* g_beta[18];
  float momentum = 0.1f;
  bool set = false;
};
thread_local BnState t_bn;

struct Ctx {
  const uncl_gen_weights* w;
  char* ws;          // workspace base for this chunk's activations
  const char* prev;  // previous frame's workspace base (video) or NULL
  Layout L;
  int n;             // tiles in this chunk
  int save_preact;   // keep pre-GELU values (training)
  int fuse_up;       // inference: up_path.3.up is recomputed inside up_path.3.conv.conv's loader (same idea, last decoder level)
  int fuse_in;       // inference: inc.conv.conv is recomputed inside inc.conv.conv1's loader (its output never goes to HBM)
  int norm;          // InstanceNorm between conv and activation (unet_norm = 'instance_norm')
  int norm_keep;     // ... and keep zhat / rstd for the backward pass
  // tiles read in place (uncl_gen_run.x_tile_off): the stack of frames, this chunk's slice of the offset table, row pitch, rows
  const float* x_frames = nullptr;
  const int32_t* x_off = nullptr;
  int x_pitch = 0, x_rows = 0;
  hipStream_t s;
  float slope() const { return w->act == UNCL_ACT_LRELU ? 0.2f : 0.f; }
  // norm + activation (+ residual) of the conv output just written to `p` (the buffer `buf`'s extent), in place
  int post_norm(void* p, int buf, const void* res = nullptr, int res_b0 = 0) const {
    const int li = norm_index(buf);
    if (li < 0) return UNCL_ERR_ARG;
    int roff = 0;
    for (int i = 0; i < li; ++i) roff += kDims[kNormBuf[i]].c;
    void* z = norm_keep ? ws + L.off[B_ZN0 + li] : nullptr;
    // 1/std: every layer owns a contiguous [N_total][C] block of the B_RSTD area; this chunk's rows start at n0
    float* rs = norm_keep ? rstd_base + (size_t)roff * n_total + (size_t)n0 * kDims[buf].c : nullptr;
    if (norm == 2) {
      // batch statistics need the whole batch in one call of this function: no chunking, activations kept (training)
      if (!t_bn.set || !norm_keep || n != n_total || n0 != 0) return UNCL_ERR_ARG;
      void* scratch = rstd_base + (size_t)RSTD_FLOATS * n_total;
      return bwd_bnorm_forward(w->dtype, p, z, rs, t_bn.gamma[li], t_bn.beta[li], t_bn.rmean[li], t_bn.rvar[li], t_bn.momentum, res,
                               res_b0, n, kDims[buf].h * kDims[buf].w, kDims[buf].c, slope(), scratch, s);
    }
    return bwd_inorm_forward(w->dtype, p, z, rs, res, res_b0, n, kDims[buf].h * kDims[buf].w, kDims[buf].c, slope(), s);
  }
  float* rstd_base;  // start of the B_RSTD area of the WHOLE call's layout
  int n_total, n0;   // tiles of the whole call, first tile of this chunk
  void* ptr(int b) const { return ws + L.off[b]; }
  Layout Lp;         // the previous frame's buffer offsets from `prev` (= L for a workspace of its own, its slice of a clip workspace)
  const void* pptr(int b) const { return prev ? prev + Lp.off[b] : nullptr; }
};

uncl_conv_desc base_desc(const Ctx& c, int wi, int ksize, int pad, int cin, int cout, int act) {
  uncl_conv_desc d = {};
  d.dtype = c.w->dtype;
  d.ksize = ksize;
  d.pad = pad;
  d.N = c.n;
  d.Cin = cin;
  d.Cout = cout;
  d.weight = c.w->w[wi];
  d.bias = c.w->b[wi];
  d.act = act;
  return d;
}

void set_src0(uncl_conv_desc& d, const Ctx& c, int b) {
  d.src0 = c.ptr(b);
  d.src0_H = kDims[b].h; d.src0_W = kDims[b].w; d.src0_C = kDims[b].c;
}
void set_out(uncl_conv_desc& d, void* p, int b) {
  d.out = p;
  d.out_H = kDims[b].h; d.out_W = kDims[b].w; d.out_C = kDims[b].c;
}

// bf16 runs the pipelined kernel (producer-side pooling); fp32 the generic one (loader-side pooling)
inline bool use_pipe(const Ctx& c) { return uncl_is_h16(c.w->dtype); }

int run3(const Ctx& c, int wi, uncl_conv_desc& d, void* pool_out) {
  ProfScope ps(wi, c.s);
  if (use_pipe(c)) return uncl_conv3x3_pipe(&d, pool_out, c.s);
  return uncl_conv_igemm(&d, c.s);
}

// 3x3 conv (valid or full) reading buffer `in` (through a 2x2 max-pool if `pool`) into buffer `out`;
// `pooled` >= 0 names the buffer that holds / receives the pooled copy in the bf16 path
int conv3(const Ctx& c, int wi, int in, int out, int cin, int cout, int pad, bool pool, int prev_ch = 0,
          int in_pooled = -1, int out_pooled = -1) {
  uncl_conv_desc d = base_desc(c, wi, 3, pad, cin, cout, c.norm ? UNCL_ACT_NONE : c.w->act);
  const bool pipe = use_pipe(c);
  const int src = (pool && pipe) ? in_pooled : in;
  set_src0(d, c, src);
  d.src_mode = (pool && !pipe) ? UNCL_SRC_MAXPOOL2 : UNCL_SRC_PLAIN;
  d.H = pool ? kDims[in].h / 2 : kDims[in].h;
  d.W = pool ? kDims[in].w / 2 : kDims[in].w;
  if (prev_ch > 0 && c.prev) { d.prev0 = c.pptr(src); d.prev_ch = prev_ch; }
  set_out(d, c.ptr(out), out);
  int rc = run3(c, wi, d, (pipe && out_pooled >= 0 && !c.norm) ? c.ptr(out_pooled) : nullptr);
  if (rc != UNCL_OK) return rc;
  if (c.norm) {
    // conv -> InstanceNorm -> activation (unet_parts.py:57-75); the pooled copy can only be taken after the norm
    if ((rc = c.post_norm(c.ptr(out), out)) != UNCL_OK) return rc;
    if (out_pooled >= 0 && (pipe || c.save_preact))
      return bwd_maxpool2(c.w->dtype, c.ptr(out), c.ptr(out_pooled), c.n, kDims[out].h, kDims[out].w, kDims[out].c, c.s);
    return UNCL_OK;
  }
  // fp32 training (parity) mode: the next stage pools inside its loader, but the backward pass reads the pooled tensor itself
  if (!pipe && out_pooled >= 0 && c.save_preact)
    return bwd_maxpool2_f32(c.ptr(out), c.ptr(out_pooled), c.n, kDims[out].h, kDims[out].w, kDims[out].c, c.s);
  return UNCL_OK;
}

// decoder stage: ConvT2x2(s2) of `x1` -> up buffer; concat-ssr(skip, up) -> ConvT3x3 -> ConvT3x3
int up_stage(const Ctx& c, int wi_up, int x1, int skip, int upbuf, int abuf, int outbuf, int ch, int cout, int prev_ch,
             void* final_out, const uncl_conv_desc* tail) {
  int rc;
  // inference, last two decoder levels: the 32- / 64-channel up-conv is recomputed per halo tile inside the concat layer's loader
  // (the 252 x 252 x 32 / 122 x 122 x 64 up-sampled map is neither written nor read back)
  const bool fuse_up = c.fuse_up && use_pipe(c) && (ch == 32 || (ch == 64 && g_fuse_up64)) && cout == 32 &&
                       2 * kDims[x1].h == kDims[skip].h && !(prev_ch > 0 && c.prev);
  if (fuse_up) {
    uncl_conv_desc d = base_desc(c, wi_up + 1, 3, 2, 4 * ch, cout, c.w->act);
    set_src0(d, c, skip);
    d.src1 = c.ptr(x1);
    d.src1_H = kDims[x1].h; d.src1_W = kDims[x1].w; d.src1_C = kDims[x1].c;
    d.src_mode = UNCL_SRC_CONCAT_SSR_UP;
    d.up_w = c.w->w[wi_up]; d.up_b = c.w->b[wi_up];
    d.H = kDims[skip].h; d.W = kDims[skip].w;
    set_out(d, c.ptr(abuf), abuf);
    if (g_fuse_tail && tail != nullptr && tail->skip_main_store && !c.norm && c.w->act == UNCL_ACT_RELU) {
      // the whole stage in one launch: neither this layer's map nor the next one's is written (only the 1-channel result)
      d.tail_w = c.w->w[wi_up + 2]; d.tail_b = c.w->b[wi_up + 2];
      d.out1_w = tail->out1_w; d.out1_b = tail->out1_b; d.out1 = tail->out1; d.out1_act = tail->out1_act;
      d.skip_main_store = 1;
      rc = run3(c, wi_up + 1, d, nullptr);
      if (rc != UNCL_ERR_ARG) return rc;
      d.tail_w = nullptr; d.tail_b = nullptr; d.out1_w = nullptr; d.out1_b = nullptr; d.out1 = nullptr; d.skip_main_store = 0;
    }
    if ((rc = run3(c, wi_up + 1, d, nullptr)) != UNCL_OK) return rc;
  } else if (use_pipe(c)) {
    ProfScope ps(wi_up, c.s);
    const int h = kDims[x1].h == 1 ? 12 : kDims[x1].h, w = kDims[x1].h == 1 ? 12 : kDims[x1].w;
    const bool pv = prev_ch > 0 && c.prev;
    if ((rc = uncl_upconv2x2_dt(c.ptr(x1), pv ? c.pptr(x1) : nullptr, pv ? prev_ch : 0, c.w->w[wi_up], c.w->b[wi_up],
                                c.ptr(upbuf), c.w->dtype, c.n, h, w, ch, ch, c.s)) != UNCL_OK)
      return rc;
  } else {
    uncl_conv_desc d = base_desc(c, wi_up, 1, 0, ch, ch, UNCL_ACT_NONE);
    set_src0(d, c, x1);
    d.src_mode = UNCL_SRC_PLAIN;
    d.H = kDims[x1].h == 1 ? 12 : kDims[x1].h;
    d.W = kDims[x1].h == 1 ? 12 : kDims[x1].w;
    if (prev_ch > 0 && c.prev) { d.prev0 = c.pptr(x1); d.prev_ch = prev_ch; }
    d.z_mode = UNCL_Z_UP2X2;
    set_out(d, c.ptr(upbuf), upbuf);
    ProfScope ps(wi_up, c.s);
    if ((rc = uncl_conv_igemm(&d, c.s)) != UNCL_OK) return rc;
  }
  if (!fuse_up) {
    uncl_conv_desc d = base_desc(c, wi_up + 1, 3, 2, 4 * ch, cout, c.norm ? UNCL_ACT_NONE : c.w->act);
    set_src0(d, c, skip);
    d.src1 = c.ptr(upbuf);
    d.src1_H = kDims[upbuf].h; d.src1_W = kDims[upbuf].w; d.src1_C = kDims[upbuf].c;
    d.src_mode = UNCL_SRC_CONCAT_SSR;
    d.H = kDims[skip].h; d.W = kDims[skip].w;
    set_out(d, c.ptr(abuf), abuf);
    if ((rc = run3(c, wi_up + 1, d, nullptr)) != UNCL_OK) return rc;
    if (c.norm && (rc = c.post_norm(c.ptr(abuf), abuf)) != UNCL_OK) return rc;
  }
  {
    uncl_conv_desc d = base_desc(c, wi_up + 2, 3, 2, cout, cout, c.norm ? UNCL_ACT_NONE : c.w->act);
    set_src0(d, c, abuf);
    d.src_mode = UNCL_SRC_PLAIN;
    d.H = kDims[abuf].h; d.W = kDims[abuf].w;
    void* outp = final_out ? final_out : c.ptr(outbuf);
    set_out(d, outp, outbuf);
    if (tail && !c.norm) {
      d.out1_w = tail->out1_w; d.out1_b = tail->out1_b; d.out1 = tail->out1; d.out1_act = tail->out1_act;
      d.skip_main_store = tail->skip_main_store;
    }
    if ((rc = run3(c, wi_up + 2, d, nullptr)) != UNCL_OK) return rc;
    if (c.norm) {
      if ((rc = c.post_norm(outp, outbuf)) != UNCL_OK) return rc;
      // the 1x1 outconv + last activation reads the normalised features (it is fused into the conv's epilogue otherwise)
      if (tail)
        return bwd_outc_forward(c.w->dtype, outp, tail->out1_w, tail->out1_b, tail->out1, (long long)c.n * 256 * 256, tail->out1_act,
                                c.s);
    }
  }
  return UNCL_OK;
}

// zpre >= 0: buffer `zpre` receives the layer's output and `out` its GELU (the training passes keep both); one launch where the
// direct 1x1 kernel takes it, else the convolution and the gelu kernel
int conv1(const Ctx& c, int wi, int in, int out, int cin, int cout, int act, const void* res, int res_b0,
          const float* scale, int groups = 0, int zpre = -1) {
  uncl_conv_desc d = base_desc(c, wi, 1, 0, groups ? cin / groups : cin, groups ? cout / groups : cout, act);
  set_src0(d, c, in);
  d.src_mode = UNCL_SRC_PLAIN;
  d.H = 12; d.W = 12;
  d.src0_H = 12; d.src0_W = 12;
  d.res = res; d.res_batch_stride0 = res_b0; d.scale_n = scale;
  if (groups) { d.z_mode = UNCL_Z_GROUPS; d.groups = groups; }
  set_out(d, c.ptr(out), out);
  d.out_H = 12; d.out_W = 12;
  if (zpre >= 0) {
    int rc = UNCL_GELU_NOT_FUSED;
    if (c.w->dtype == UNCL_BF16) rc = uncl_conv1x1_gelu(&d, c.ptr(zpre), 1, c.s);
    if (rc != UNCL_GELU_NOT_FUSED) return rc;
    set_out(d, c.ptr(zpre), zpre);
    d.out_H = 12; d.out_W = 12;
    if ((rc = uncl_conv_igemm(&d, c.s)) != UNCL_OK) return rc;
    return bwd_gelu_forward(c.w->dtype, c.ptr(zpre), c.ptr(out), (long long)c.n * NODES * cout, c.s);
  }
  return uncl_conv_igemm(&d, c.s);
}

// phase 0: the whole network; 1: everything up to the third decoder stage; 2: the last decoder stage only
// `segs`: which segments of the network this call runs -- SEG_A the 252 ... 57 pixel encoder levels (inc, down_path.0 / 1), SEG_B the
// 28 ... 12 pixel levels and their way back up (down_path.2 / 3, graph block, up_path.0 / 1: launches that are latency-bound or
// fill a fraction of the chip at any batch size), SEG_C up_path.2, SEG_D the last decoder stage.  A multi-stream forward runs
// different segments at different granularities (uncl_gen_forward).
enum { SEG_A = 1, SEG_B = 2, SEG_C = 4, SEG_D = 8, SEG_ALL = 15 };
int run_chunk(const Ctx& c, const float* x, float* out, void* up_x, int32_t* knn_out, const float* drop0,
              const float* drop1, int segs = SEG_ALL) {
  const uncl_gen_weights* w = c.w;
  int rc;
#define RUN(e) do { if ((rc = (e)) != UNCL_OK) return rc; } while (0)
  if (segs & SEG_A) {
  // encoder
  if (c.fuse_in) {
    uncl_conv_desc d = base_desc(c, W_INC1, 3, 0, 32, 32, w->act);
    d.src_mode = UNCL_SRC_IMAGE1;
    d.src0 = x; d.src0_H = S_IN; d.src0_W = S_IN; d.src0_C = 1;
    if (c.x_off != nullptr) { d.src0 = c.x_frames; d.src1 = c.x_off; d.src1_W = c.x_pitch; d.src1_H = c.x_rows; }
    d.pre_w = w->inc0_w; d.pre_b = w->inc0_b;
    d.H = S_INC0; d.W = S_INC0;
    set_out(d, c.ptr(B_X0), B_X0);
    RUN(run3(c, W_INC1, d, c.ptr(B_X0P)));
  } else {
    if (c.x_off != nullptr) return UNCL_ERR_ARG;       // tiles in place need the fused first layer: the caller gathers instead
    RUN(uncl_conv_in_c1(x, w->inc0_w, w->inc0_b, c.ptr(B_INC0), w->dtype, c.n, S_IN, S_IN, 32, c.norm ? UNCL_ACT_NONE : w->act, c.s));
    if (c.norm) RUN(c.post_norm(c.ptr(B_INC0), B_INC0));
    RUN(conv3(c, W_INC1, B_INC0, B_X0, 32, 32, 0, false, 0, -1, B_X0P));
  }
  RUN(conv3(c, W_D0A, B_X0, B_D0A, 32, 64, 0, true, 1, B_X0P));
  RUN(conv3(c, W_D0B, B_D0A, B_X1, 64, 64, 0, false, 0, -1, B_X1P));
  RUN(conv3(c, W_D1A, B_X1, B_D1A, 64, 128, 0, true, 2, B_X1P));
  RUN(conv3(c, W_D1B, B_D1A, B_X2, 128, 128, 0, false, 0, -1, B_X2P));
  }
  if (segs & SEG_B) {
  RUN(conv3(c, W_D2A, B_X2, B_D2A, 128, 256, 0, true, 4, B_X2P));
  RUN(conv3(c, W_D2B, B_D2A, B_X3, 256, 256, 0, false, 0, -1, B_X3P));
  RUN(conv3(c, W_D3A, B_X3, B_D3A, 256, 256, 0, true, 8, B_X3P));
  {
    // transposed 3x3 back to 12x12, ReLU, then + pos_embed (Unet_singleFrame.py:94) fused as a broadcast residual
    uncl_conv_desc d = base_desc(c, W_D3B, 3, 2, 256, 256, c.norm ? UNCL_ACT_NONE : w->act);
    set_src0(d, c, B_D3A);
    d.H = S_D3A; d.W = S_D3A;
    if (!c.norm) { d.res = w->pos_embed; d.res_batch_stride0 = 1; }
    set_out(d, c.ptr(B_X4), B_X4);
    RUN(run3(c, W_D3B, d, nullptr));
    if (c.norm) RUN(c.post_norm(c.ptr(B_X4), B_X4, w->pos_embed, 1));
  }
  // graph block: Grapher (fc1 -> kNN -> max-relative -> grouped 1x1 + GELU -> fc2, residual) then FFN
  int32_t* knn = reinterpret_cast<int32_t*>(c.ptr(B_KNN));
  const bool fused = g_fused_graph && uncl_is_h16(w->dtype) && !c.save_preact && drop0 == nullptr && drop1 == nullptr;
  if (fused && g_fused_graph == 1) {
    // inference: the whole block is one launch with the intermediates in LDS (csrc/graph_block.hip)
    RUN(uncl_gcn_block(c.ptr(B_X4), w->w[W_GFC1], w->b[W_GFC1], w->relative_pos, w->w[W_GGC], w->b[W_GGC], w->w[W_GFC2],
                       w->b[W_GFC2], w->w[W_FFC1], w->b[W_FFC1], w->w[W_FFC2], w->b[W_FFC2], knn, c.ptr(B_GOUT), w->dtype, c.n,
                       c.s));
    if (knn_out && hipMemcpyAsync(knn_out, knn, (size_t)c.n * NODES * 9 * 4, hipMemcpyDeviceToDevice, c.s) != hipSuccess)
      return UNCL_ERR_LAUNCH;
  } else {
    RUN(conv1(c, W_GFC1, B_X4, B_GFC1, 256, 256, UNCL_ACT_NONE, nullptr, 0, nullptr));
    RUN(uncl_gcn_knn(c.ptr(B_GFC1), w->dtype, w->relative_pos, knn, nullptr, c.n, NODES, 256, 9, nullptr, c.s));
    if (knn_out && hipMemcpyAsync(knn_out, knn, (size_t)c.n * NODES * 9 * 4, hipMemcpyDeviceToDevice, c.s) != hipSuccess)
      return UNCL_ERR_LAUNCH;
    if (fused) {
      // g_fused_graph == 2: fc1 and the kNN as their own launches, the rest of the block fused (A/B)
      RUN(uncl_gcn_tail(c.ptr(B_GFC1), knn, c.ptr(B_X4), w->w[W_GGC], w->b[W_GGC], w->w[W_GFC2], w->b[W_GFC2], w->w[W_FFC1],
                        w->b[W_FFC1], w->w[W_FFC2], w->b[W_FFC2], c.ptr(B_GOUT), w->dtype, c.n, c.s));
    } else {
      RUN(uncl_gcn_maxrel(c.ptr(B_GFC1), knn, c.ptr(B_GMR), w->dtype, c.n, NODES, 256, 9, c.s));
      if (c.save_preact) {
        RUN(conv1(c, W_GGC, B_GMR, B_GGC, 512, 512, UNCL_ACT_NONE, nullptr, 0, nullptr, 4, B_GGCZ));
      } else {
        RUN(conv1(c, W_GGC, B_GMR, B_GGC, 512, 512, UNCL_ACT_GELU, nullptr, 0, nullptr, 4));
      }
      RUN(conv1(c, W_GFC2, B_GGC, B_GX1, 512, 256, UNCL_ACT_NONE, c.ptr(B_X4), 0, drop0));
      if (c.save_preact) {
        RUN(conv1(c, W_FFC1, B_GX1, B_FH, 256, 256, UNCL_ACT_NONE, nullptr, 0, nullptr, 0, B_FHZ));
      } else {
        RUN(conv1(c, W_FFC1, B_GX1, B_FH, 256, 256, UNCL_ACT_GELU, nullptr, 0, nullptr));
      }
      RUN(conv1(c, W_FFC2, B_FH, B_GOUT, 256, 256, UNCL_ACT_NONE, c.ptr(B_GX1), 0, drop1));
    }
  }
  // decoder
  RUN(up_stage(c, W_U0UP, B_GOUT, B_X3, B_U0UP, B_U0A, B_U0, 256, 128, 8, nullptr, nullptr));
  RUN(up_stage(c, W_U1UP, B_U0, B_X2, B_U1UP, B_U1A, B_U1, 128, 64, 4, nullptr, nullptr));
  }
  if (segs & SEG_C) RUN(up_stage(c, W_U2UP, B_U1, B_X1, B_U2UP, B_U2A, B_U2, 64, 32, 2, nullptr, nullptr));
  if (!(segs & SEG_D)) return UNCL_OK;
  uncl_conv_desc tail = {};
  tail.out1_w = w->outc_w; tail.out1_b = w->outc_b; tail.out1 = out; tail.out1_act = w->last_act;
  tail.skip_main_store = up_x == nullptr ? 1 : 0;
  RUN(up_stage(c, W_U3UP, B_U2, B_X0, B_U3UP, B_U3A, B_UPX, 32, 32, 1, (c.norm && !up_x) ? c.ptr(B_UPX) : up_x, &tail));
#undef RUN
  return UNCL_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// Backward pass (bf16).  G(b) = gradient buffer of activation buffer b (same layout in the gradient workspace);
// every stored gradient already carries the activation derivative of the layer that produced b.
// ------------------------------------------------------------------------------------------------------------------
struct BwdScratch {
  char* tmp;     // fp32 mode only, (N,252,252,128): a data gradient before its masked / accumulating store
  char* gcat;    // (N,252,252,128) bf16: gradient of a decoder stage's concat input
  char* gpool;   // (N,126,126,32) bf16: gradient of a pooled encoder input
  char* tA;      // (N,144,512) bf16 x 4 temporaries of the graph block
  char* tB;
  char* tC;
  char* tD;
  float* f32;    // (N,144,256) fp32 max-relative scatter target
  char* mix;     // (N,126,126,32) bf16: a stage input with the previous frame's head channels (video)
  float* misc;   // outc / conv_in partial sums
  float* cs;     // staged column sums of the bias gradients (512 floats per channel, CS_CHANNELS channels)
};
constexpr int CS_CHANNELS = 8192;

// bias gradients of one backward pass: stage 1 per layer, one finishing launch at the end
struct ColsumQueue {
  uncl_colsum_item it[UNCL_COLSUM_MAX_ITEMS];
  int n = 0;
  int channels = 0;
};

// carry arena of the recurrent hand-off: head-channel gradients of the eight mixed stage inputs, bf16 (N, pixels, C/32)
struct CarrySlot { int buf, pix, pc; };
const CarrySlot kCarry[8] = {{B_X0P, 126 * 126, 1}, {B_X1P, 61 * 61, 2}, {B_X2P, 28 * 28, 4}, {B_X3P, 12 * 12, 8},
                             {B_GOUT, 144, 8},      {B_U0, 28 * 28, 4},  {B_U1, 61 * 61, 2},  {B_U2, 126 * 126, 1}};
size_t carry_off(int slot, int N, size_t es = 2) {
  size_t o = 0;
  for (int i = 0; i < slot; ++i) o += ((size_t)N * kCarry[i].pix * kCarry[i].pc * es + 255) & ~(size_t)255;
  return o;
}

size_t bwd_scratch_bytes(int N, size_t es = 2) {
  size_t b = 0;
  if (es == 4) b += (size_t)N * 252 * 252 * 128 * 4;     // tmp
  b += (size_t)N * 252 * 252 * 128 * es;
  b += (size_t)N * 126 * 126 * 32 * es;
  b += 4 * (size_t)N * 144 * 512 * es;
  b += (size_t)N * 144 * 256 * 4;
  b += (size_t)N * 126 * 126 * 32 * es;
  b += ((size_t)1024 * 33 + 64 + (size_t)512 * 320 + 320 + (size_t)512 * 512) * 4;
  b += (size_t)CS_CHANNELS * 512 * 4;
  if (es == 2) b += 2 * uncl_wgrad_scratch_bytes();      // per-group partial sums of the deterministic weight gradients (two streams)
  return b + 4096;
}

#ifndef UNCL_BWD_WSTREAM_DEFAULT
#define UNCL_BWD_WSTREAM_DEFAULT 1
#endif
// one weight-gradient stream (+ its fork / join events) per device, created under a lock on first use; at the other priority
// level, like the forward's side streams: its hardware queue is then never the caller's
// `pass`: held by a backward pass for as long as it issues work on the stream -- the fork / join events are one pair per device, so
// two host threads running passes on one device must not interleave their event records (a wait would bind to the other thread's
// record: a missing dependency); they take turns enqueueing instead, the GPU work still overlaps
struct WgradStream { hipStream_t s = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; std::mutex* pass = nullptr; };
static std::mutex g_wgs_mu;
static std::map<int, WgradStream> g_wgs;
static WgradStream* wgrad_stream_for_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_wgs_mu);
  auto it = g_wgs.find(dev);
  if (it != g_wgs.end()) return &it->second;
  WgradStream w;
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  const hipError_t e = greatest != least ? hipStreamCreateWithPriority(&w.s, hipStreamNonBlocking, greatest)
                                         : hipStreamCreateWithFlags(&w.s, hipStreamNonBlocking);
  if (e != hipSuccess || hipEventCreateWithFlags(&w.ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&w.ev_join, hipEventDisableTiming) != hipSuccess)
    return nullptr;
  w.pass = new std::mutex();      // lives as long as the process, like the stream
  return &(g_wgs[dev] = w);
}

struct BCtx {
  const uncl_gen_weights* w;
  const uncl_gen_bwd* b;
  char* fws;   // forward workspace
  const char* pws;  // previous frame's forward workspace (video, frame > 0) or NULL
  char* gws;   // gradient arena (same layout)
  Layout L;
  Layout Lp;     // the previous frame's buffer offsets from `pws` (= L for a workspace of its own; its slice of a clip workspace)
  bool defer = false;   // clip layout: this call leaves the 3x3 / 2x2 weight and bias gradients to the clip's last call
  BwdScratch sc;
  int n;
  int dt;        // UNCL_BF16 (training) or UNCL_F32 (parity mode: deterministic plain-fp32 kernels, bwd_f32.hip)
  size_t es;
  float slope;
  bool fused_bias = false;   // bf16: 3x3 layers' bias gradients come out of the weight-gradient kernel (atomics into gb)
  hipStream_t s;
  // weight-gradient stream (image passes in bf16, UNCL_BWD_WSTREAM): the weight / bias gradients of a layer depend on nothing
  // later in the pass and nothing depends on them before the pass ends, so they run beside the data-gradient chain and fill
  // the ramp-down of its launches.  wfork(): everything issued on `s` so far is ordered before what follows on `ws`.
  hipStream_t ws = nullptr;
  hipEvent_t ev_wfork = nullptr, ev_wjoin = nullptr;
  // deterministic weight gradients (bf16): scratch for their per-group partial sums, one half per stream they can be launched
  // on (the graph block's 1x1 gradients stay on the caller's stream, everything else goes to `ws` when it exists)
  hipStream_t main_s = nullptr;
  char* wdet = nullptr;
  size_t wdet_half = 0;
  void wdet_select() const {
    if (wdet != nullptr) (void)uncl_wgrad_set_scratch(wdet + (s != main_s ? wdet_half : 0), wdet_half);
  }
  hipStream_t wfork() const {
    if (!ws) return s;
    if (hipEventRecord(ev_wfork, s) != hipSuccess || hipStreamWaitEvent(ws, ev_wfork, 0) != hipSuccess) return s;
    return ws;
  }
  int wjoin() const {
    if (!ws) return UNCL_OK;
    if (hipEventRecord(ev_wjoin, ws) != hipSuccess || hipStreamWaitEvent(s, ev_wjoin, 0) != hipSuccess) return UNCL_ERR_LAUNCH;
    return UNCL_OK;
  }
  ColsumQueue* q;
  void* F(int buf) const { return fws + L.off[buf]; }
  void* G(int buf) const { return gws + L.off[buf]; }
  // out[c] (+)= sum_rows x[row][c], finished by flush_colsums()
  int colsum(const void* x, long long rows, int C, float* out) const {
    if (dt == UNCL_F32) return bwd_colsum_f32(x, rows, C, C, out, b->accumulate, s);
    for (int c0 = 0; c0 < C; c0 += 256) {
      const int cw = C - c0 < 256 ? C - c0 : 256;
      if (q->n >= UNCL_COLSUM_MAX_ITEMS || q->channels + cw > CS_CHANNELS) return UNCL_ERR_ARG;
      const int rc = uncl_colsum_bf16_stage((const bf16_t*)x + c0, rows, cw, C, sc.cs + (size_t)q->channels * 512, out + c0,
                                            b->accumulate, &q->it[q->n], s);
      if (rc != UNCL_OK) return rc;
      q->n += 1; q->channels += cw;
    }
    return UNCL_OK;
  }
  int flush_colsums() const {
    if (wjoin() != UNCL_OK) return UNCL_ERR_LAUNCH;      // the staged sums (and the weight gradients) come from the side stream
    const int rc = uncl_colsum_finish(q->it, q->n, s);
    q->n = 0; q->channels = 0;
    return rc;
  }
  // InstanceNorm models: dL/dzhat (what arrives, already masked by the activation derivative) -> dL/dz of the conv output
  // that activation buffer `buf` holds, in place, before that conv's weight / data gradients are taken
  Layout LZ;     // the forward workspace's layout incl. the zhat / rstd buffers
  int unnorm(void* g, int buf) const {
    if (!w->norm) return UNCL_OK;
    const int li = norm_index(buf);
    if (li < 0) return UNCL_ERR_ARG;
    int roff = 0;
    for (int i = 0; i < li; ++i) roff += kDims[kNormBuf[i]].c;
    const float* rs = reinterpret_cast<const float*>(fws + LZ.off[B_RSTD]) + (size_t)roff * n;
    if (w->norm == 2) {
      if (!t_bn.set) return UNCL_ERR_ARG;
      void* scratch = reinterpret_cast<float*>(fws + LZ.off[B_RSTD]) + (size_t)RSTD_FLOATS * n;      // the forward workspace's scratch tail
      return bwd_bnorm_backward(dt, g, fws + LZ.off[B_ZN0 + li], rs, t_bn.gamma[li], t_bn.g_gamma[li], t_bn.g_beta[li], b->accumulate, n,
                                kDims[buf].h * kDims[buf].w, kDims[buf].c, scratch, s);
    }
    return bwd_inorm_backward(dt, g, fws + LZ.off[B_ZN0 + li], rs, n, kDims[buf].h * kDims[buf].w, kDims[buf].c, s);
  }
  bool video() const { return b->carry_in != nullptr || b->carry_out != nullptr; }
  const void* cin(int slot) const { return b->carry_in ? (const char*)b->carry_in + carry_off(slot, n, es) : nullptr; }
  void* cout(int slot) const { return b->carry_out ? (char*)b->carry_out + carry_off(slot, n, es) : nullptr; }
  // the tensor a hand-off stage actually read: this frame's buffer with the previous frame's head channels
  const void* mixed(int slot) const {
    const CarrySlot& k = kCarry[slot];
    if (!pws) return F(k.buf);
    if (bwd_mix_heads(dt, F(k.buf), pws + Lp.off[k.buf], sc.mix, (long long)n * k.pix, kDims[k.buf].c, k.pc, s) != UNCL_OK) return nullptr;
    return sc.mix;
  }
  int handoff(int slot, void* g, const void* mask) const {
    if (!video()) return UNCL_OK;
    const CarrySlot& k = kCarry[slot];
    return bwd_head_handoff(dt, g, mask, slope, cin(slot), cout(slot), (long long)n * k.pix, kDims[k.buf].c, k.pc, s);
  }
  // an encoder stage's hand-off (on the gradient of its pooled input, no mask) and the max-pool backward that reads it: one launch
  // for clips (UNCL_POOL_HANDOFF=0: two, for A/B and the bit-for-bit test)
  int pool_handoff(int slot, const void* g_pool, const void* x, void* G_x, int H, int W, int C) const {
    static const int fuse = [] { const char* e = getenv("UNCL_POOL_HANDOFF"); return e ? atoi(e) : 1; }();
    const CarrySlot& k = kCarry[slot];
    if (!video() || !fuse || k.pix != (H / 2) * (W / 2) || kDims[k.buf].c != C) {
      const int rc = handoff(slot, const_cast<void*>(g_pool), nullptr);
      if (rc != UNCL_OK) return rc;
      return bwd_pool_backward(dt, g_pool, x, G_x, n, H, W, C, slope, 1, s);
    }
    return bwd_pool_backward_handoff(dt, g_pool, x, G_x, n, H, W, C, slope, 1, cin(slot), cout(slot), k.pc, s);
  }
};

uncl_conv_desc bdesc(const BCtx& c, int ks, int pad, int h, int w, int cin, int cout) {
  uncl_conv_desc d = {};
  d.dtype = c.dt; d.ksize = ks; d.pad = pad; d.N = c.n; d.H = h; d.W = w; d.Cin = cin; d.Cout = cout;
  d.src_mode = UNCL_SRC_PLAIN; d.act = UNCL_ACT_NONE;
  return d;
}

// weight gradient in the pass's element type: matrix-core kernel with fp32 atomics (bf16) or the deterministic fp32 kernel
int conv_wgrad(const BCtx& c, const uncl_conv_desc& d, const void* gy, float* gw) {
  if (c.dt == UNCL_F32) return bwd_wgrad_f32(&d, gy, gw, c.s);
  c.wdet_select();
  return uncl_conv_wgrad(&d, gy, gw, c.s);
}
// weight AND bias gradient of a 3x3 layer (output gradient gy of oh x ow x cout): bf16 passes take the bias sums from the
// weight-gradient kernel's own pass over gy (atomics into gb: zeroed at the start of a non-accumulating pass, see
// uncl_gen_backward); fp32 passes keep the deterministic column-sum kernels
int conv_wgrad_bias(const BCtx& c, const uncl_conv_desc& d, const void* gy, float* gw, float* gb, int oh, int ow, int cout) {
  if (c.dt == UNCL_BF16 && c.fused_bias) {
    c.wdet_select();
    return uncl_conv_wgrad_bias(&d, gy, gw, gb, c.s);
  }
  const int rc = conv_wgrad(c, d, gy, gw);
  if (rc != UNCL_OK) return rc;
  return c.colsum(gy, (long long)c.n * oh * ow, cout, gb);
}

// weight + bias gradient of a 3x3 layer whose input is buffer `xin` (plain) and whose output gradient is gy
int wgrad3(const BCtx& c, int wi, int xin, int pad, int cin, int cout, const void* gy, int oh, int ow,
           const void* src = nullptr) {
  if (c.defer) return UNCL_OK;
  uncl_conv_desc d = bdesc(c, 3, pad, kDims[xin].h, kDims[xin].w, cin, cout);
  d.src0 = src ? src : c.F(xin); d.src0_H = kDims[xin].h; d.src0_W = kDims[xin].w; d.src0_C = kDims[xin].c;
  BCtx cw = c;
  cw.s = c.wfork();
  return conv_wgrad_bias(cw, d, gy, c.b->gw[wi], c.b->gb[wi], oh, ow, cout);
}

// data gradient of a 3x3 layer: gy (N,gh,gw,gc) -> out buffer (cout_d channels), pad_d = 2 - pad_fwd
int dgrad3(const BCtx& c, int wi, const void* gy, int gh, int gw, int gc, int pad_d, int cout_d, void* out, int oh, int ow,
           const void* mask, int accumulate) {
  uncl_conv_desc d = bdesc(c, 3, pad_d, gh, gw, gc, cout_d);
  d.src0 = gy; d.src0_H = gh; d.src0_W = gw; d.src0_C = gc;
  d.weight = c.b->wd[wi];
  d.out = out; d.out_H = oh; d.out_W = ow; d.out_C = cout_d;
  if (c.dt == UNCL_F32) {
    // exact-fp32 implicit GEMM, then the gradient-mode store (ReLU mask of the producing layer / accumulation) as its own pass
    const bool post = mask != nullptr || accumulate;
    if (post) d.out = c.sc.tmp;
    const int rc = uncl_conv_igemm(&d, c.s);
    if (rc != UNCL_OK || !post) return rc;
    return bwd_mask_acc_f32(c.sc.tmp, mask, c.slope, out, accumulate, (long long)c.n * oh * ow * cout_d, c.s);
  }
  return uncl_conv3x3_dgrad(&d, mask, c.slope, accumulate, c.s);
}

// 1x1 conv on the 144-node graph tensors: weight gradient / data gradient
int wgrad1(const BCtx& c, int wi, const void* x, int xc_total, int cin, const void* gy, int gy_total, int cout, float* gw,
           bool bias) {
  if (c.defer) return UNCL_OK;
  uncl_conv_desc d = bdesc(c, 1, 0, 12, 12, cin, cout);
  d.src0 = x; d.src0_H = 12; d.src0_W = 12; d.src0_C = xc_total;
  d.out_C = gy_total;  // leading dimension of gy
  int rc = conv_wgrad(c, d, gy, gw);
  if (rc != UNCL_OK || !bias) return rc;
  return c.colsum(gy, (long long)c.n * NODES, gy_total, c.b->gb[wi]);
}
// gelu_z != NULL: the result is multiplied by gelu'(gelu_z) (the pre-activation this gradient flows back through), in the same
// launch where the direct 1x1 kernel takes it
int dgrad1(const BCtx& c, int wi, const void* gy, int cin_d, int cout_d, void* out, const void* res, int groups = 0,
           const void* gelu_z = nullptr) {
  uncl_conv_desc d = bdesc(c, 1, 0, 12, 12, groups ? cin_d / groups : cin_d, groups ? cout_d / groups : cout_d);
  d.src0 = gy; d.src0_H = 12; d.src0_W = 12; d.src0_C = cin_d;
  d.weight = c.b->wd[wi];
  d.res = res;
  d.out = out; d.out_H = 12; d.out_W = 12; d.out_C = cout_d;
  if (groups) { d.z_mode = UNCL_Z_GROUPS; d.groups = groups; }
  if (gelu_z != nullptr) {
    int rc = UNCL_GELU_NOT_FUSED;
    if (c.dt == UNCL_BF16) rc = uncl_conv1x1_gelu(&d, const_cast<void*>(gelu_z), 2, c.s);
    if (rc != UNCL_GELU_NOT_FUSED) return rc;
    if ((rc = uncl_conv_igemm(&d, c.s)) != UNCL_OK) return rc;
    return bwd_gelu_backward(c.dt, out, gelu_z, out, (long long)c.n * 144 * cout_d, c.s);
  }
  return uncl_conv_igemm(&d, c.s);
}

int backward_all(const BCtx& c) {
  const uncl_gen_bwd* b = c.b;
  int rc;
#define RUN(e) do { if ((rc = (e)) != UNCL_OK) return rc; } while (0)
  // ---- tail: outconv + sigmoid
  RUN(bwd_outc_backward(c.dt, b->g_out, b->x_out, b->g_upx, b->up_x, c.w->outc_w, c.G(B_UPX), b->g_outc_w, b->g_outc_b,
                         (long long)c.n * 256 * 256, c.w->last_act, c.slope, b->accumulate, c.sc.misc, c.s));
  // ---- decoder stages 3..0
  struct Stage { int wi, x1, skip, up, a, out, ch, cout; };
  const Stage st[4] = {{W_U0UP, B_GOUT, B_X3, B_U0UP, B_U0A, B_U0, 256, 128},
                       {W_U1UP, B_U0, B_X2, B_U1UP, B_U1A, B_U1, 128, 64},
                       {W_U2UP, B_U1, B_X1, B_U2UP, B_U2A, B_U2, 64, 32},
                       {W_U3UP, B_U2, B_X0, B_U3UP, B_U3A, B_UPX, 32, 32}};
  for (int i = 3; i >= 0; --i) {
    const Stage& q = st[i];
    const int oh = kDims[q.out].h, ow = kDims[q.out].w, ah = kDims[q.a].h, aw = kDims[q.a].w;
    const int sh = kDims[q.skip].h, sw = kDims[q.skip].w, uh = kDims[q.up].h, uw = kDims[q.up].w;
    // conv b (ConvT 3x3, cout -> cout): input = buffer a
    RUN(c.unnorm(c.G(q.out), q.out));
    RUN(wgrad3(c, q.wi + 2, q.a, 2, q.cout, q.cout, c.G(q.out), oh, ow));
    RUN(dgrad3(c, q.wi + 2, c.G(q.out), oh, ow, q.cout, 0, q.cout, c.G(q.a), ah, aw, c.F(q.a), 0));
    // conv a (ConvT 3x3 on the concat, 4ch -> cout)
    RUN(c.unnorm(c.G(q.a), q.a));
    if (!c.defer) {
      uncl_conv_desc d = bdesc(c, 3, 2, sh, sw, 4 * q.ch, q.cout);
      d.src_mode = UNCL_SRC_CONCAT_SSR;
      d.src0 = c.F(q.skip); d.src0_H = sh; d.src0_W = sw; d.src0_C = q.ch;
      d.src1 = c.F(q.up); d.src1_H = uh; d.src1_W = uw; d.src1_C = q.ch;
      BCtx cw = c;
      cw.s = c.wfork();
      // (rounds 3 - 4a kept a column-sum launch for the two stages whose weight gradient ran in the old 64 x 64 kernel, which had no
      // registers left for the bias sums; the four-member kernel that takes every skip-concat layer now sums them itself)
      RUN(conv_wgrad_bias(cw, d, c.G(q.a), b->gw[q.wi + 1], b->gb[q.wi + 1], ah, aw, q.cout));
    }
    if ((b->ssr_fused >> i) & 1) {
      // the skip operator's backward as the data-gradient launch's epilogue: the (N, H, W, 4 ch) gradient of the concatenation is
      // neither written nor read back (the caller packed this layer's data-gradient weights in the interleaved cout order)
      if (c.dt != UNCL_BF16 || sh != uh || sw != uw) return UNCL_ERR_ARG;
      uncl_conv_desc d = bdesc(c, 3, 0, ah, aw, q.cout, 4 * q.ch);
      d.src0 = c.G(q.a); d.src0_H = ah; d.src0_W = aw; d.src0_C = q.cout;
      d.weight = b->wd[q.wi + 1];
      d.out = nullptr; d.out_H = sh; d.out_W = sw; d.out_C = 4 * q.ch;
      RUN(uncl_conv3x3_dgrad_ssr(&d, c.F(q.skip), c.G(q.skip), c.G(q.up), c.slope, 0, c.s));
    } else {
      RUN(dgrad3(c, q.wi + 1, c.G(q.a), ah, aw, q.cout, 0, 4 * q.ch, c.sc.gcat, sh, sw, nullptr, 0));
      RUN(bwd_ssr_backward(c.dt, c.sc.gcat, c.F(q.skip), c.G(q.skip), c.G(q.up), c.n, sh, sw, q.ch, uh, uw, c.slope, 0, c.s));
    }
    // up (ConvT 2x2 s2, ch -> ch): input x1
    const int xh = kDims[q.x1].h == 1 ? 12 : kDims[q.x1].h, xw = kDims[q.x1].h == 1 ? 12 : kDims[q.x1].w;
    const int slot = 4 + i;  // hand-off slot of this stage's input (video): GOUT, U0, U1, U2
    const void* x1m = c.defer ? nullptr : c.mixed(slot);
    if (!x1m && !c.defer) return UNCL_ERR_LAUNCH;
    if (!c.defer) {
      BCtx cw = c;
      cw.s = c.wfork();
      if (c.dt == UNCL_F32) RUN(bwd_upconv2x2_wgrad_f32(x1m, c.G(q.up), b->gw[q.wi], c.n, xh, xw, q.ch, q.ch, cw.s));
      else {
        cw.wdet_select();
        RUN(uncl_upconv2x2_wgrad(x1m, c.G(q.up), b->gw[q.wi], c.n, xh, xw, q.ch, q.ch, cw.s));
      }
      RUN(cw.colsum(c.G(q.up), (long long)c.n * uh * uw, q.ch, b->gb[q.wi]));
    }
    // the ReLU derivative of the layer that produced x1 is applied by the dgrad kernel (single frames) or, for clips,
    // by the hand-off kernel after the head channels have been exchanged between frames
    const void* x1mask = i == 0 ? nullptr : c.F(q.x1);
    if (c.dt == UNCL_F32)
      RUN(bwd_upconv2x2_dgrad_f32(c.G(q.up), b->wd[q.wi], c.video() ? nullptr : x1mask, c.slope, c.G(q.x1), c.n, xh, xw, q.ch, q.ch,
                                  c.s));
    else {
      // clips: the hand-off of this stage's input gradient rides in the launch's store (UNCL_UP_HANDOFF=0: its own launch)
      static const int fuse = [] { const char* e = getenv("UNCL_UP_HANDOFF"); return e ? atoi(e) : 1; }();
      const CarrySlot& ks = kCarry[slot];
      if (c.video() && fuse && ks.pix == xh * xw && kDims[ks.buf].c == q.ch) {
        RUN(bwd_upconv2x2_dgrad_handoff(c.G(q.up), b->wd[q.wi], x1mask, c.slope, c.G(q.x1), c.n, xh, xw, q.ch, q.ch, c.cin(slot),
                                        c.cout(slot), ks.pc, c.s));
        continue;
      }
      RUN(uncl_upconv2x2_dgrad(c.G(q.up), b->wd[q.wi], c.video() ? nullptr : x1mask, c.slope, c.G(q.x1), c.n, xh, xw, q.ch, q.ch,
                               c.s));
    }
    RUN(c.handoff(slot, c.G(q.x1), x1mask));
  }
  // the decoder's parameter gradients are final from here on (biases: staged column sums are flushed first)
  if (b->ev_decoder_done && !c.defer) {
    RUN(c.flush_colsums());
    if (hipEventRecord(reinterpret_cast<hipEvent_t>(b->ev_decoder_done), c.s) != hipSuccess) return UNCL_ERR_LAUNCH;
  }
  // ---- graph block
  const float* drop0 = b->drop_scale;
  const float* drop1 = b->drop_scale ? b->drop_scale + c.n : nullptr;
  const long long per256 = (long long)NODES * 256, per512 = (long long)NODES * 512;
  // The five output gradients the 1x1 weight gradients read are scratch in a single pass; a clip pass (deferred weight gradients)
  // parks them in arena slots of the same width that the backward pass does not use otherwise -- G(FH), G(FHZ), G(GOUT) (free once
  // its last reader, the residual of fc1's data gradient, has run), G(GGCZ), G(GFC1) -- where the clip's last call finds all frames
  void* tA1 = c.defer ? c.G(B_FH) : static_cast<void*>(c.sc.tA);
  void* tB1 = c.defer ? c.G(B_FHZ) : static_cast<void*>(c.sc.tB);
  void* tA2 = c.defer ? c.G(B_GOUT) : static_cast<void*>(c.sc.tA);
  void* tC = c.defer ? c.G(B_GGCZ) : static_cast<void*>(c.sc.tC);
  void* tB2 = c.defer ? c.G(B_GFC1) : static_cast<void*>(c.sc.tB);
  // FFN: GOUT = drop1 * fc2(gelu(fc1(GX1))) + GX1
  RUN(bwd_scale_rows(c.dt, c.G(B_GOUT), drop1, tA1, c.n, per256, c.s));
  RUN(wgrad1(c, W_FFC2, c.F(B_FH), 256, 256, tA1, 256, 256, b->gw[W_FFC2], true));
  RUN(dgrad1(c, W_FFC2, tA1, 256, 256, tB1, nullptr, 0, c.F(B_FHZ)));
  RUN(wgrad1(c, W_FFC1, c.F(B_GX1), 256, 256, tB1, 256, 256, b->gw[W_FFC1], true));
  RUN(dgrad1(c, W_FFC1, tB1, 256, 256, c.G(B_GX1), c.G(B_GOUT)));
  // Grapher: GX1 = drop0 * fc2(gelu(gconv(maxrel(fc1(X4))))) + X4
  RUN(bwd_scale_rows(c.dt, c.G(B_GX1), drop0, tA2, c.n, per256, c.s));
  RUN(wgrad1(c, W_GFC2, c.F(B_GGC), 512, 512, tA2, 256, 256, b->gw[W_GFC2], true));
  RUN(dgrad1(c, W_GFC2, tA2, 256, 512, tC, nullptr, 0, c.F(B_GGCZ)));
  if (!c.defer) {  // grouped 1x1: four independent 128 -> 128 blocks, one launch (grid.y = group)
    uncl_conv_desc d = bdesc(c, 1, 0, 12, 12, 128, 128);
    d.src0 = c.F(B_GMR); d.src0_H = 12; d.src0_W = 12; d.src0_C = 512;
    d.out_C = 512;
    d.z_mode = UNCL_Z_GROUPS; d.groups = 4;
    RUN(conv_wgrad(c, d, tC, b->gw[W_GGC]));
    RUN(c.colsum(tC, (long long)c.n * NODES, 512, b->gb[W_GGC]));
  }
  RUN(dgrad1(c, W_GGC, tC, 512, 512, c.sc.tD, nullptr, 4));
  RUN(bwd_gcn_maxrel_backward(c.dt, c.sc.tD, c.F(B_GFC1), reinterpret_cast<const int32_t*>(c.F(B_KNN)), c.sc.f32, tB2, c.n, NODES,
                               256, 9, c.s));
  RUN(wgrad1(c, W_GFC1, c.F(B_X4), 256, 256, tB2, 256, 256, b->gw[W_GFC1], true));
  RUN(dgrad1(c, W_GFC1, tB2, 256, 256, c.G(B_X4), c.G(B_GX1)));
  RUN(bwd_sum_samples(c.dt, c.G(B_X4), b->g_pos_embed, c.n, per256, b->accumulate, c.s));
  // (deferred weight gradients: the masked gradient replaces G(X4) element by element, so that it is still there at the clip's end)
  void* g4 = c.defer ? c.G(B_X4) : static_cast<void*>(c.sc.tA);
  RUN(bwd_mask_minus(c.dt, c.G(B_X4), c.F(B_X4), c.w->pos_embed, g4, c.n, per256, c.slope, c.s));
  // ---- encoder
  // down3: conv (valid, pooled X3 -> D3A), ConvT (D3A -> X4)
  RUN(c.unnorm(g4, B_X4));
  RUN(wgrad3(c, W_D3B, B_D3A, 2, 256, 256, g4, 12, 12));
  RUN(dgrad3(c, W_D3B, g4, 12, 12, 256, 0, 256, c.G(B_D3A), 10, 10, c.F(B_D3A), 0));
  struct Enc { int wa, wb, xin, pooled, mid, out, cin, cout; };
  // second conv of each level first (its output gradient is complete), then the first conv + pool backward
  const Enc en[3] = {{W_D2A, W_D2B, B_X2, B_X2P, B_D2A, B_X3, 128, 256},
                     {W_D1A, W_D1B, B_X1, B_X1P, B_D1A, B_X2, 64, 128},
                     {W_D0A, W_D0B, B_X0, B_X0P, B_D0A, B_X1, 32, 64}};
  // down3's first conv reads pooled X3
  RUN(c.unnorm(c.G(B_D3A), B_D3A));
  if (!c.defer) {
    const void* xm = c.mixed(3);
    if (!xm) return UNCL_ERR_LAUNCH;
    RUN(wgrad3(c, W_D3A, B_X3P, 0, 256, 256, c.G(B_D3A), 10, 10, xm));
  }
  RUN(dgrad3(c, W_D3A, c.G(B_D3A), 10, 10, 256, 2, 256, c.sc.gpool, 12, 12, nullptr, 0));
  RUN(c.pool_handoff(3, c.sc.gpool, c.F(B_X3), c.G(B_X3), 24, 24, 256));
  for (int i = 0; i < 3; ++i) {
    const Enc& e = en[i];
    const int oh = kDims[e.out].h, mh = kDims[e.mid].h, ph = kDims[e.pooled].h, xh = kDims[e.xin].h;
    RUN(c.unnorm(c.G(e.out), e.out));
    RUN(wgrad3(c, e.wb, e.mid, 0, e.cout, e.cout, c.G(e.out), oh, oh));
    RUN(dgrad3(c, e.wb, c.G(e.out), oh, oh, e.cout, 2, e.cout, c.G(e.mid), mh, mh, c.F(e.mid), 0));
    const void* xm = c.defer ? nullptr : c.mixed(2 - i);
    if (!xm && !c.defer) return UNCL_ERR_LAUNCH;
    RUN(c.unnorm(c.G(e.mid), e.mid));
    RUN(wgrad3(c, e.wa, e.pooled, 0, e.cin, e.cout, c.G(e.mid), mh, mh, xm));
    RUN(dgrad3(c, e.wa, c.G(e.mid), mh, mh, e.cout, 2, e.cin, c.sc.gpool, ph, ph, nullptr, 0));
    RUN(c.pool_handoff(2 - i, c.sc.gpool, c.F(e.xin), c.G(e.xin), xh, xh, e.cin));
  }
  // inc: conv1 (INC0 -> X0), conv (image -> INC0)
  RUN(c.unnorm(c.G(B_X0), B_X0));
  RUN(wgrad3(c, W_INC1, B_INC0, 0, 32, 32, c.G(B_X0), 252, 252));
  RUN(dgrad3(c, W_INC1, c.G(B_X0), 252, 252, 32, 2, 32, c.G(B_INC0), 254, 254, c.F(B_INC0), 0));
  RUN(c.unnorm(c.G(B_INC0), B_INC0));
  // (clip passes: once per clip, see deferred_wgrads)
  if (!c.defer)
    RUN(bwd_conv_in_c1_wgrad(c.dt, c.G(B_INC0), b->x, b->g_inc0_w, b->g_inc0_b, c.n, 256, 256, b->accumulate, c.sc.misc, c.s));
#undef RUN
  return UNCL_OK;
}

// The 3x3 / 2x2 weight and bias gradients of a whole clip in one launch per layer (uncl_gen_bwd.clip_T): `c` spans all
// T * nf samples of the clip workspace and of the gradient arena (c.n = T * nf, offsets of frame 0), whose slices the frames'
// data-gradient chains have filled.  A hand-off stage's input is its frame's buffer with the head channels of the frame
// before it (frame 0: its own), rebuilt for the whole clip in `mix`.
int deferred_wgrads(const BCtx& c, int nf, int T) {
  const uncl_gen_bwd* b = c.b;
  int rc;
#define RUN(e) do { if ((rc = (e)) != UNCL_OK) return rc; } while (0)
  auto mixed_all = [&](int slot) -> const void* {
    const CarrySlot& k = kCarry[slot];
    if (T == 1) return c.F(k.buf);
    if (bwd_mix_heads_clip(c.dt, c.F(k.buf), c.sc.mix, (long long)T * nf * k.pix, (long long)nf * k.pix, kDims[k.buf].c, k.pc, c.s) !=
        UNCL_OK)
      return nullptr;
    return c.sc.mix;
  };
  struct Stage { int wi, x1, skip, up, a, out, ch, cout; };
  const Stage st[4] = {{W_U0UP, B_GOUT, B_X3, B_U0UP, B_U0A, B_U0, 256, 128},
                       {W_U1UP, B_U0, B_X2, B_U1UP, B_U1A, B_U1, 128, 64},
                       {W_U2UP, B_U1, B_X1, B_U2UP, B_U2A, B_U2, 64, 32},
                       {W_U3UP, B_U2, B_X0, B_U3UP, B_U3A, B_UPX, 32, 32}};
  for (int i = 3; i >= 0; --i) {
    const Stage& q = st[i];
    const int oh = kDims[q.out].h, ow = kDims[q.out].w, ah = kDims[q.a].h, aw = kDims[q.a].w;
    const int sh = kDims[q.skip].h, sw = kDims[q.skip].w, uh = kDims[q.up].h, uw = kDims[q.up].w;
    RUN(wgrad3(c, q.wi + 2, q.a, 2, q.cout, q.cout, c.G(q.out), oh, ow));
    {
      uncl_conv_desc d = bdesc(c, 3, 2, sh, sw, 4 * q.ch, q.cout);
      d.src_mode = UNCL_SRC_CONCAT_SSR;
      d.src0 = c.F(q.skip); d.src0_H = sh; d.src0_W = sw; d.src0_C = q.ch;
      d.src1 = c.F(q.up); d.src1_H = uh; d.src1_W = uw; d.src1_C = q.ch;
      RUN(conv_wgrad_bias(c, d, c.G(q.a), b->gw[q.wi + 1], b->gb[q.wi + 1], ah, aw, q.cout));
    }
    const int xh = kDims[q.x1].h == 1 ? 12 : kDims[q.x1].h, xw = kDims[q.x1].h == 1 ? 12 : kDims[q.x1].w;
    const void* x1m = mixed_all(4 + i);
    if (!x1m) return UNCL_ERR_LAUNCH;
    if (c.dt == UNCL_F32) RUN(bwd_upconv2x2_wgrad_f32(x1m, c.G(q.up), b->gw[q.wi], c.n, xh, xw, q.ch, q.ch, c.s));
    else {
      c.wdet_select();
      RUN(uncl_upconv2x2_wgrad(x1m, c.G(q.up), b->gw[q.wi], c.n, xh, xw, q.ch, q.ch, c.s));
    }
    RUN(c.colsum(c.G(q.up), (long long)c.n * uh * uw, q.ch, b->gb[q.wi]));
  }
  if (b->ev_decoder_done) {
    RUN(c.flush_colsums());
    if (hipEventRecord(reinterpret_cast<hipEvent_t>(b->ev_decoder_done), c.s) != hipSuccess) return UNCL_ERR_LAUNCH;
  }
  // the graph block's five 1x1 layers (their output gradients were parked in the arena by every frame, see backward_all)
  RUN(wgrad1(c, W_FFC2, c.F(B_FH), 256, 256, c.G(B_FH), 256, 256, b->gw[W_FFC2], true));
  RUN(wgrad1(c, W_FFC1, c.F(B_GX1), 256, 256, c.G(B_FHZ), 256, 256, b->gw[W_FFC1], true));
  RUN(wgrad1(c, W_GFC2, c.F(B_GGC), 512, 512, c.G(B_GOUT), 256, 256, b->gw[W_GFC2], true));
  {
    uncl_conv_desc d = bdesc(c, 1, 0, 12, 12, 128, 128);
    d.src0 = c.F(B_GMR); d.src0_H = 12; d.src0_W = 12; d.src0_C = 512;
    d.out_C = 512;
    d.z_mode = UNCL_Z_GROUPS; d.groups = 4;
    RUN(conv_wgrad(c, d, c.G(B_GGCZ), b->gw[W_GGC]));
    RUN(c.colsum(c.G(B_GGCZ), (long long)c.n * NODES, 512, b->gb[W_GGC]));
  }
  RUN(wgrad1(c, W_GFC1, c.F(B_X4), 256, 256, c.G(B_GFC1), 256, 256, b->gw[W_GFC1], true));
  RUN(wgrad3(c, W_D3B, B_D3A, 2, 256, 256, c.G(B_X4), 12, 12));
  {
    const void* xm = mixed_all(3);
    if (!xm) return UNCL_ERR_LAUNCH;
    RUN(wgrad3(c, W_D3A, B_X3P, 0, 256, 256, c.G(B_D3A), 10, 10, xm));
  }
  struct Enc { int wa, wb, pooled, mid, out, cin, cout; };
  const Enc en[3] = {{W_D2A, W_D2B, B_X2P, B_D2A, B_X3, 128, 256},
                     {W_D1A, W_D1B, B_X1P, B_D1A, B_X2, 64, 128},
                     {W_D0A, W_D0B, B_X0P, B_D0A, B_X1, 32, 64}};
  for (int i = 0; i < 3; ++i) {
    const Enc& e = en[i];
    const int oh = kDims[e.out].h, mh = kDims[e.mid].h;
    RUN(wgrad3(c, e.wb, e.mid, 0, e.cout, e.cout, c.G(e.out), oh, oh));
    const void* xm = mixed_all(2 - i);
    if (!xm) return UNCL_ERR_LAUNCH;
    RUN(wgrad3(c, e.wa, e.pooled, 0, e.cin, e.cout, c.G(e.mid), mh, mh, xm));
  }
  RUN(wgrad3(c, W_INC1, B_INC0, 0, 32, 32, c.G(B_X0), 252, 252));
  // first layer (1 -> 32 channels): frame 0's `x` is the start of the clip's (T * nf, 256, 256) input array (uncl_gen_bwd.clip_T);
  // nothing else writes these two slots in a clip pass, so they are overwritten, not added to
  RUN(bwd_conv_in_c1_wgrad(c.dt, c.G(B_INC0), b->x, b->g_inc0_w, b->g_inc0_b, c.n, 256, 256, 0, c.sc.misc, c.s));
#undef RUN
  return UNCL_OK;
}

}  // namespace

extern "C" int uncl_gen_set_fused_graph(int on) {
  const int old = g_fused_graph;
  g_fused_graph = on < 0 ? 0 : (on > 2 ? 2 : on);
  return old;
}

// streams an un-chunked inference batch of >= 64 tiles is spread over (1 = the caller's stream only), see uncl_gen_forward
// hybrid schedule (uncl_gen_forward): 0 off; P > 0: the small-map segment in P parts on P streams, everything else once for the batch
static int g_hybrid = [] { const char* e = getenv("UNCL_HYBRID"); return e ? atoi(e) : UNCL_HYBRID_DEFAULT; }();
static int g_streams = 2;     // measured (every conv a one-workgroup-per-CU producer/consumer launch): 1 / 2 / 4 streams = 4.86 / 4.67 / 4.78 ms
extern "C" int uncl_gen_set_streams(int n) {
  if (n < 1 || n > 4) return UNCL_ERR_ARG;
  g_streams = n;
  return UNCL_OK;
}

extern "C" int uncl_prof_enable(int layer, int max_records) {
  for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
  delete[] g_prof.ev;
  g_prof = Prof();
  if (layer < 0 || max_records <= 0) return UNCL_OK;
  g_prof.ev = new hipEvent_t[2 * max_records];
  for (int i = 0; i < 2 * max_records; ++i)
    if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return UNCL_ERR_LAUNCH;
  g_prof.layer = layer;
  g_prof.cap = max_records;
  return UNCL_OK;
}

extern "C" int uncl_prof_read(float* ms_host, int max_n) {
  int n = g_prof.count < max_n ? g_prof.count : max_n;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return UNCL_ERR_LAUNCH;
    if (hipEventElapsedTime(&ms_host[i], g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return UNCL_ERR_LAUNCH;
  }
  g_prof.count = 0;
  return n;
}

extern "C" const char* uncl_gen_layer_name(int i) {
  if (i < 0 || i >= UNCL_G_NUM_WEIGHTS) return nullptr;
  return kNames[i];
}

extern "C" size_t uncl_gen_workspace_bytes(int N, int chunk, int dtype, int keep_activations) {
  return uncl_gen_workspace_bytes_ex(N, chunk, dtype, keep_activations, 0);
}
extern "C" size_t uncl_gen_workspace_bytes_ex(int N, int chunk, int dtype, int keep_activations, int norm) {
  if (N <= 0) return 0;
  if (chunk <= 0 || chunk > N) chunk = N;
  return make_layout(keep_activations ? N : chunk, dtype, norm && keep_activations).total;
}

// Side streams and fork / join events of the multi-stream forward, one set per device (a process may drive several GPUs, and
// two host threads may call uncl_gen_forward at once: creation is under a lock; calls on ONE device from several threads
// share the set and must be serialised by the caller, like any use of one stream).
struct SideStreams {
  static constexpr int MAX_SIDE = 3;
  hipStream_t side[MAX_SIDE] = {};
  hipEvent_t ev_fork = nullptr, ev_fork2 = nullptr, ev_join[MAX_SIDE] = {};
};
static std::mutex g_side_mu;
static std::map<int, SideStreams> g_side;
static SideStreams* side_streams_for_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_side_mu);
  auto it = g_side.find(dev);
  if (it != g_side.end()) return &it->second;
  SideStreams ss;
  if (hipEventCreateWithFlags(&ss.ev_fork, hipEventDisableTiming) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&ss.ev_fork2, hipEventDisableTiming) != hipSuccess) return nullptr;
  // The runtime deals its hardware queues (GPU_MAX_HW_QUEUES, default 4) to streams round-robin PER PRIORITY LEVEL: a
  // normal-priority side stream created after three other streams (a process that has initialised RCCL has them) lands on the
  // caller's queue and the two parts of the forward serialise -- measured 4.68 instead of 4.25 ms per step, exactly the
  // one-queue time.  Side streams of another priority level come from another pool, whatever was created before them.
  static const int prio_on = [] { const char* e = getenv("UNCL_SIDE_PRIORITY"); return e ? atoi(e) : 1; }();
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  for (int i = 0; i < SideStreams::MAX_SIDE; ++i) {
    const hipError_t e = (prio_on && greatest != least)
                             ? hipStreamCreateWithPriority(&ss.side[i], hipStreamNonBlocking, greatest)
                             : hipStreamCreateWithFlags(&ss.side[i], hipStreamNonBlocking);
    if (e != hipSuccess || hipEventCreateWithFlags(&ss.ev_join[i], hipEventDisableTiming) != hipSuccess) return nullptr;
  }
  return &(g_side[dev] = ss);
}

extern "C" int uncl_gen_forward(const uncl_gen_weights* w, const uncl_gen_run* r, void* stream) {
  if (!w || !r || !r->x || !r->out || !r->workspace || r->N <= 0) return UNCL_ERR_ARG;
  if (w->dtype != UNCL_F32 && !uncl_is_h16(w->dtype)) return UNCL_ERR_ARG;
  // fp16 is the inference dtype (BASELINE configs[4]): no kept activations for a backward pass, no clip recurrence
  if (w->dtype == UNCL_F16 && (r->keep_activations || r->save_preact || r->prev_workspace != nullptr)) return UNCL_ERR_ARG;
  // tiles read in place: only where the first layer is rebuilt in the second layer's loader (the conditions of Ctx::fuse_in below)
  if (r->x_tile_off != nullptr && (r->x_pitch < 256 || r->x_rows < 256 || !uncl_is_h16(w->dtype) || r->keep_activations || r->save_preact ||
                                   r->prev_workspace != nullptr || w->norm != 0))
    return UNCL_ERR_ARG;
  int chunk = r->chunk;
  if (chunk <= 0 || chunk > r->N) chunk = r->N;
  // clip layout: one workspace for clip_T frames of N samples, this call is frame clip_t (see uncl_gen_run)
  const bool clip = r->clip_T > 0;
  if (clip && (r->clip_t < 0 || r->clip_t >= r->clip_T || !r->keep_activations || chunk != r->N || w->norm != 0 ||
               r->prev_workspace != nullptr))
    return UNCL_ERR_ARG;
  const int n_alloc = clip ? r->clip_T * r->N : (r->keep_activations ? r->N : chunk);
  if (w->norm < 0 || w->norm > 2) return UNCL_ERR_ARG;
  // batch_norm (2) is a TRAINING mode of this entry (eval folds the running statistics into the weights on the host): the whole
  // batch in one chunk, activations kept, the layers' parameters announced with uncl_gen_set_bn
  if (w->norm == 2 && (!r->keep_activations || chunk != r->N || !t_bn.set)) return UNCL_ERR_ARG;
  Layout L = make_layout(n_alloc, w->dtype, w->norm && r->keep_activations);
  if (r->workspace_bytes < L.total) return UNCL_ERR_ARG;
  const size_t es = uncl_is_h16(w->dtype) ? 2 : 4;
  // A large un-chunked inference batch runs as up to four contiguous parts on as many streams (the caller's and internal
  // ones, joined by events): the launches of one part fill the ramp-down of the others' persistent grids and the gaps
  // between dependent launches.  Every buffer is (N, ...), so the parts own disjoint slices of the same workspace.
  const int split_cfg = g_hybrid > 0 ? (g_hybrid < SideStreams::MAX_SIDE + 1 ? g_hybrid : SideStreams::MAX_SIDE + 1) : g_streams;
  const bool split2 = split_cfg >= 2 && chunk == r->N && r->N >= 64 && !r->keep_activations && !r->save_preact &&
                      r->prev_workspace == nullptr;
  constexpr int MAX_SIDE = SideStreams::MAX_SIDE;
  hipStream_t main_s = reinterpret_cast<hipStream_t>(stream);
  // at least 32 tiles per part
  const int parts = split2 ? (r->N / 32 < split_cfg ? r->N / 32 : split_cfg) : 1;
  SideStreams* ss = nullptr;
  if (split2) {
    ss = side_streams_for_current_device();     // one set per device, created under a lock on first use
    if (!ss) return UNCL_ERR_LAUNCH;
    if (hipEventRecord(ss->ev_fork, main_s) != hipSuccess) return UNCL_ERR_LAUNCH;
    for (int i = 0; i < parts - 1; ++i)
      if (hipStreamWaitEvent(ss->side[i], ss->ev_fork, 0) != hipSuccess) return UNCL_ERR_LAUNCH;
    chunk = (r->N + parts - 1) / parts;
  }
  // Two parts of UNEQUAL size (round 5): with equal halves both streams reach the bottleneck block -- ten dependent launches of
  // 30 - 110 us that fill a fraction of the chip whatever the batch -- at the same time, and for ~0.35 ms of the step only small
  // launches run; with a 60 : 40 split the smaller part is through it while the larger one still runs its encoder.
  // UNCL_SPLIT_PCT: per cent of the tiles in the first part (50: equal halves)
  static const int split_pct = [] { const char* e = getenv("UNCL_SPLIT_PCT"); const int v = e ? atoi(e) : UNCL_SPLIT_PCT_DEFAULT; return v < 20 ? 20 : (v > 80 ? 80 : v); }();
  int first_chunk = chunk;
  if (split2 && parts == 2 && split_pct != 50) {
    first_chunk = (int)((long long)r->N * split_pct / 100);
    if (first_chunk < 32) first_chunk = 32;
    if (r->N - first_chunk < 32) first_chunk = r->N - 32;
  }
  // whatever happens after the fork, the caller's stream waits for the side streams before this call returns: the caller may
  // free the workspace as soon as its own stream is done
  auto join_sides = [&]() {
    int rc = UNCL_OK;
    if (!ss) return rc;
    for (int i = 0; i < parts - 1; ++i)
      if (hipEventRecord(ss->ev_join[i], ss->side[i]) != hipSuccess || hipStreamWaitEvent(main_s, ss->ev_join[i], 0) != hipSuccess)
        rc = UNCL_ERR_LAUNCH;
    return rc;
  };
  // HYBRID schedule (round 5, UNCL_HYBRID / uncl_gen_set_streams(-P)): the big-map segments run ONCE for the whole batch on the
  // caller's stream (they fill the chip by themselves, and a launch of 200 tiles has a smaller tail than two of 100), only the
  // small-map segment -- whose launches are latency-bound or fill a fraction of the chip whatever the batch -- runs as P parts on
  // P streams.  Same arithmetic per tile: results are bit-identical to every other schedule.
  if (split2 && g_hybrid > 0 && !clip) {
    auto ctx_for = [&](int n0, int n, hipStream_t st) {
      Ctx c;
      c.w = w;
      c.n = n;
      c.save_preact = r->save_preact;
      c.fuse_in = uncl_is_h16(w->dtype) && !r->keep_activations && !r->save_preact && r->prev_workspace == nullptr && !w->norm;
      c.fuse_up = c.fuse_in;
      c.norm = w->norm;
      c.norm_keep = w->norm && r->keep_activations;
      c.rstd_base = reinterpret_cast<float*>(reinterpret_cast<char*>(r->workspace) + L.off[B_RSTD]);
      c.n_total = r->N; c.n0 = n0;
      c.x_frames = r->x; c.x_off = r->x_tile_off ? r->x_tile_off + n0 : nullptr; c.x_pitch = r->x_pitch; c.x_rows = r->x_rows;
      c.s = st;
      Layout Lc = L;
      for (int b = 0; b < B_COUNT; ++b) Lc.off[b] = L.off[b] + L.per_n[b] * (size_t)n0;
      c.L = Lc;
      c.Lp = Lc;
      c.ws = reinterpret_cast<char*>(r->workspace);
      c.prev = nullptr;
      return c;
    };
    const Ctx whole = ctx_for(0, r->N, main_s);
    int rc = run_chunk(whole, r->x, r->out, r->up_x, nullptr, nullptr, nullptr, SEG_A);
    if (rc != UNCL_OK) { (void)join_sides(); return rc; }
    int P = g_hybrid < MAX_SIDE + 1 ? g_hybrid : MAX_SIDE + 1;
    if (r->N / 32 < P) P = r->N / 32;
    if (P < 1) P = 1;
    const int pc = (r->N + P - 1) / P;
    // (the side streams waited for the fork event recorded BEFORE segment A: they need segment A's results -> a second fork)
    if (hipEventRecord(ss->ev_fork2, main_s) != hipSuccess) return UNCL_ERR_LAUNCH;
    for (int i = 0; i < P - 1; ++i)
      if (hipStreamWaitEvent(ss->side[i], ss->ev_fork2, 0) != hipSuccess) return UNCL_ERR_LAUNCH;
    for (int pi = 0, n0 = 0; n0 < r->N; ++pi, n0 += pc) {
      const int n = r->N - n0 < pc ? r->N - n0 : pc;
      const Ctx cp = ctx_for(n0, n, pi == 0 ? main_s : ss->side[pi - 1]);
      rc = run_chunk(cp, r->x + (size_t)n0 * 256 * 256, r->out + (size_t)n0 * 256 * 256, nullptr,
                     r->knn_idx ? r->knn_idx + (size_t)n0 * NODES * 9 : nullptr, r->drop_scale ? r->drop_scale + n0 : nullptr,
                     r->drop_scale ? r->drop_scale + r->N + n0 : nullptr, SEG_B);
      if (rc != UNCL_OK) { (void)join_sides(); return rc; }
    }
    // join every side stream that took a part (join_sides covers parts - 1 of them: the same or more)
    for (int i = 0; i < P - 1; ++i)
      if (hipEventRecord(ss->ev_join[i], ss->side[i]) != hipSuccess || hipStreamWaitEvent(main_s, ss->ev_join[i], 0) != hipSuccess)
        return UNCL_ERR_LAUNCH;
    return run_chunk(whole, r->x, r->out, r->up_x, nullptr, nullptr, nullptr, SEG_C | SEG_D);
  }
  int part_idx = 0;
  for (int n0 = 0, this_chunk = first_chunk; n0 < r->N; n0 += this_chunk, this_chunk = (split2 && parts == 2) ? r->N - first_chunk : chunk, ++part_idx) {
    Ctx c;
    c.w = w;
    c.L = L;
    c.n = (r->N - n0 < this_chunk) ? r->N - n0 : this_chunk;
    c.save_preact = r->save_preact;
    // the backward pass reads inc.conv.conv's output (ReLU mask, weight gradient), the video path hands its channels on
    // a norm sits between the fused layers' convolution and activation: no loader-side recomputation then
    c.fuse_in = uncl_is_h16(w->dtype) && !r->keep_activations && !r->save_preact && r->prev_workspace == nullptr && !w->norm;
    c.fuse_up = c.fuse_in;
    c.norm = w->norm;
    c.norm_keep = w->norm && r->keep_activations;
    c.rstd_base = reinterpret_cast<float*>(reinterpret_cast<char*>(r->workspace) + L.off[B_RSTD]);
    c.n_total = r->N; c.n0 = n0;
    c.x_frames = r->x; c.x_off = r->x_tile_off ? r->x_tile_off + n0 : nullptr; c.x_pitch = r->x_pitch; c.x_rows = r->x_rows;
    c.s = (split2 && n0 > 0) ? ss->side[part_idx - 1] : main_s;
    // with keep_activations every tile owns its slice of each buffer; otherwise the chunk's slices are reused.
    // Buffers are addressed per tile, so a chunk at tile offset n0 starts per_n*n0 bytes into each buffer.
    Layout Lc = L;
    if (r->keep_activations || split2)
      for (int b = 0; b < B_COUNT; ++b) Lc.off[b] = L.off[b] + L.per_n[b] * (size_t)n0;
    c.L = Lc;
    c.Lp = Lc;
    c.ws = reinterpret_cast<char*>(r->workspace);
    c.prev = reinterpret_cast<const char*>(r->prev_workspace);
    if (clip) {
      for (int b = 0; b < B_COUNT; ++b) {
        c.L.off[b] = L.off[b] + L.per_n[b] * (size_t)r->clip_t * r->N;
        c.Lp.off[b] = L.off[b] + L.per_n[b] * (size_t)(r->clip_t > 0 ? r->clip_t - 1 : 0) * r->N;
      }
      c.prev = r->clip_t > 0 ? c.ws : nullptr;
    }
    if (c.prev && !r->keep_activations) { (void)join_sides(); return UNCL_ERR_ARG; }
    void* up = r->up_x ? reinterpret_cast<char*>(r->up_x) + (size_t)n0 * 256 * 256 * 32 * es : nullptr;
    // split mode: the halves run everything up to the third decoder stage; the last stage (a quarter of the step in its two
    // largest launches, which fill the chip on their own) then runs once for the whole batch on the caller's stream
    // (same-box A/B: +0.3 ... +0.7 % per step against splitting it too, and the dominant kernel runs undisturbed)
    static const int tail_whole_on = [] { const char* e = getenv("UNCL_TAIL_WHOLE"); return e ? atoi(e) : 1; }();   // 0: A/B, every part runs its own last stage
    const bool tail_whole = split2 && tail_whole_on;
    // (A staggered start -- the second part gated on an event behind the first part's 252 ... 57 pixel encoder levels, so that each
    // part crosses the latency-bound bottleneck block under the other's large launches -- was measured in rounds 5 and 6 and lost both
    // times, 3.83 -> 3.98 ms: the large launches are persistent one-workgroup-per-CU grids that hold every CU for 100 - 250 us, and a
    // part's small dependent launches then wait for CUs instead of slipping in: DESIGN.md section 3.2.  Also measured in round 6
    // and removed: the bottleneck block proper (down_path.3 + graph block) ONCE for the whole batch between a join and a second fork of
    // the halves -- 3.81 -> 3.88 ms: two more synchronisation points cost more than the block's launches gain from running alone.)
    int rc = run_chunk(c, r->x + (size_t)n0 * 256 * 256, r->out + (size_t)n0 * 256 * 256, up,
                       r->knn_idx ? r->knn_idx + (size_t)n0 * NODES * 9 : nullptr,
                       r->drop_scale ? r->drop_scale + n0 : nullptr,
                       r->drop_scale ? r->drop_scale + r->N + n0 : nullptr, tail_whole ? (SEG_A | SEG_B | SEG_C) : SEG_ALL);
    if (rc != UNCL_OK) { (void)join_sides(); return rc; }
    if (tail_whole && n0 + this_chunk >= r->N) {
      // join, then the last decoder stage for the whole batch on the caller's stream
      if (join_sides() != UNCL_OK) return UNCL_ERR_LAUNCH;
      Ctx cw = c;
      cw.n = r->N; cw.L = L; cw.s = main_s; cw.n0 = 0;
      return run_chunk(cw, r->x, r->out, r->up_x, nullptr, nullptr, nullptr, SEG_D);
    }
  }
  return UNCL_OK;
}

// unet_norm = 'batch_norm', training mode (w->norm == 2): the eighteen BatchNorm2d layers in the order inc.conv.norm, inc.conv.norm1,
// down_path.0..3 (norm, norm1), up_path.0..3 (norm, norm1) -- fp32 device pointers: weight, bias, running_mean, running_var (updated
// in place by the forward), and the gradient slots of weight / bias (written by the backward; NULL arrays for a forward-only
// caller).  Thread-local; stays set until called again (all-NULL arrays clear it).
extern "C" int uncl_gen_set_bn(const float* const* gamma, const float* const* beta, float* const* running_mean,
                               float* const* running_var, float momentum, float* const* g_gamma, float* const* g_beta) {
  if (gamma == nullptr || beta == nullptr) { t_bn.set = false; return UNCL_OK; }
  for (int i = 0; i < 18; ++i) {
    if (gamma[i] == nullptr || beta[i] == nullptr) return UNCL_ERR_ARG;
    t_bn.gamma[i] = gamma[i]; t_bn.beta[i] = beta[i];
    t_bn.rmean[i] = running_mean ? running_mean[i] : nullptr;
    t_bn.rvar[i] = running_var ? running_var[i] : nullptr;
    t_bn.g_gamma[i] = g_gamma ? g_gamma[i] : nullptr;
    t_bn.g_beta[i] = g_beta ? g_beta[i] : nullptr;
  }
  t_bn.momentum = momentum;
  t_bn.set = true;
  return UNCL_OK;
}

extern "C" size_t uncl_gen_backward_workspace_bytes(int N) { return uncl_gen_backward_workspace_bytes_dt(N, UNCL_BF16); }
extern "C" size_t uncl_gen_backward_workspace_bytes_dt(int N, int dtype) {
  if (N <= 0 || (dtype != UNCL_BF16 && dtype != UNCL_F32)) return 0;
  return make_layout(N, dtype).total + bwd_scratch_bytes(N, dtype == UNCL_F32 ? 4 : 2);
}

extern "C" size_t uncl_gen_carry_bytes(int N) { return uncl_gen_carry_bytes_dt(N, UNCL_BF16); }
extern "C" size_t uncl_gen_carry_bytes_dt(int N, int dtype) {
  if (N <= 0) return 0;
  return carry_off(8, N, dtype == UNCL_F32 ? 4 : 2);
}

extern "C" int uncl_gen_backward(const uncl_gen_weights* w, const uncl_gen_bwd* b, void* stream) {
  if (!w || !b || (w->dtype != UNCL_BF16 && w->dtype != UNCL_F32) || b->N <= 0) return UNCL_ERR_ARG;
  if (!b->x || !b->x_out || !b->g_out || !b->up_x || !b->workspace || !b->grad_workspace) return UNCL_ERR_ARG;
  // clip layout (see uncl_gen_bwd): both arenas span clip_T * N samples, this call is frame clip_t and defers the 3x3 / 2x2 weight
  // gradients to the call for frame 0
  const bool clip = b->clip_T > 0;
  if (clip && (b->clip_t < 0 || b->clip_t >= b->clip_T || w->norm != 0 || b->prev_workspace != nullptr)) return UNCL_ERR_ARG;
  const int NA = clip ? b->clip_T * b->N : b->N;     // samples the arenas are laid out for
  if (b->grad_workspace_bytes < uncl_gen_backward_workspace_bytes_dt(NA, w->dtype)) return UNCL_ERR_ARG;
  for (int i = 0; i < UNCL_G_NUM_WEIGHTS; ++i)
    if (!b->wd[i] || !b->gw[i] || !b->gb[i]) return UNCL_ERR_ARG;
  if (!b->g_inc0_w || !b->g_inc0_b || !b->g_outc_w || !b->g_outc_b || !b->g_pos_embed) return UNCL_ERR_ARG;
  if ((clip ? b->clip_t > 0 : b->prev_workspace != nullptr) != (b->carry_out != nullptr)) return UNCL_ERR_ARG;
  BCtx c;
  c.w = w; c.b = b; c.n = b->N;
  c.dt = w->dtype;
  c.es = w->dtype == UNCL_F32 ? 4 : 2;
  c.L = make_layout(NA, w->dtype);
  c.LZ = make_layout(NA, w->dtype, w->norm);
  c.fws = reinterpret_cast<char*>(b->workspace);
  c.pws = reinterpret_cast<const char*>(b->prev_workspace);
  c.gws = reinterpret_cast<char*>(b->grad_workspace);
  c.slope = w->act == UNCL_ACT_LRELU ? 0.2f : 0.f;
  c.s = reinterpret_cast<hipStream_t>(stream);
  char* p = c.gws + c.L.total;
  const Layout Lall = c.L;
  c.Lp = c.L;
  if (clip) {
    for (int i = 0; i < B_COUNT; ++i) {
      c.L.off[i] = Lall.off[i] + Lall.per_n[i] * (size_t)b->clip_t * b->N;
      c.Lp.off[i] = Lall.off[i] + Lall.per_n[i] * (size_t)(b->clip_t > 0 ? b->clip_t - 1 : 0) * b->N;
    }
    c.pws = b->clip_t > 0 ? c.fws : nullptr;
    c.defer = true;
  }
  const int N = NA;     // the scratch areas are carved for the arena's sample count (`mix` holds a whole clip in the deferred pass)
  const size_t es = c.es;
  c.sc.tmp = nullptr;
  if (es == 4) { c.sc.tmp = p; p += (size_t)N * 252 * 252 * 128 * 4; }
  c.sc.gcat = p; p += (size_t)N * 252 * 252 * 128 * es;
  c.sc.gpool = p; p += (size_t)N * 126 * 126 * 32 * es;
  c.sc.tA = p; p += (size_t)N * 144 * 512 * es;
  c.sc.tB = p; p += (size_t)N * 144 * 512 * es;
  c.sc.tC = p; p += (size_t)N * 144 * 512 * es;
  c.sc.tD = p; p += (size_t)N * 144 * 512 * es;
  c.sc.f32 = reinterpret_cast<float*>(p); p += (size_t)N * 144 * 256 * 4;
  c.sc.mix = p; p += (size_t)N * 126 * 126 * 32 * es;
  c.sc.misc = reinterpret_cast<float*>(p); p += ((size_t)1024 * 33 + 64 + (size_t)512 * 320 + 320 + (size_t)512 * 512) * 4;
  c.sc.cs = reinterpret_cast<float*>(p); p += (size_t)CS_CHANNELS * 512 * 4;
  // bf16 passes in deterministic mode (uncl_gen_set_deterministic): weight / bias gradients as per-group partial sums + a
  // fixed-order reduction, the max-relative scatter in its gather form -- no float atomics, two passes over the same inputs give
  // the same bits.  Off by default: same box, N = 32 image step, 8.23 ms against 7.87 with atomics (43 more small launches and
  // ~0.6 GB of partial sums written and read back per step)
  const int det_on = g_bwd_det.load(std::memory_order_relaxed);
  struct ScratchGuard {
    bool on;
    explicit ScratchGuard(bool o) : on(o) {}
    ~ScratchGuard() { if (on) (void)uncl_wgrad_set_scratch(nullptr, 0); }
  } scratch_guard(det_on && es == 2);
  c.main_s = c.s;
  if (det_on && es == 2) { c.wdet = p; c.wdet_half = uncl_wgrad_scratch_bytes(); }
  ColsumQueue q;
  c.q = &q;
  // weight gradients beside the data-gradient chain: single-frame bf16 passes (a clip's passes share scratch between frames:
  // `mix`, the carries), see BCtx::wfork
  static const int wstream_on = [] { const char* e = getenv("UNCL_BWD_WSTREAM"); return e ? atoi(e) : UNCL_BWD_WSTREAM_DEFAULT; }();
  std::unique_lock<std::mutex> pass_lock;     // taken below iff this pass uses the device's weight-gradient stream; released on return
  // ... and not while the caller's stream is being captured: a replayed hipGraph pays ~0.2 ms per cross-stream edge on this
  // runtime (the N = 32 step: 7.9 ms on one stream, 17 ms with the ~45 forks of this pass captured; eager: 7.83 -> 7.60 ms)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(c.s, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
  if (wstream_on && cap == hipStreamCaptureStatusNone && c.dt == UNCL_BF16 && !c.video() && c.pws == nullptr && !clip) {
    WgradStream* wsd = wgrad_stream_for_current_device();
    if (!wsd) return UNCL_ERR_LAUNCH;
    c.ws = wsd->s; c.ev_wfork = wsd->ev_fork; c.ev_wjoin = wsd->ev_join;
    pass_lock = std::unique_lock<std::mutex>(*wsd->pass);
  }
  // bias gradients of the 3x3 layers out of the weight-gradient kernels (bf16): they ADD into gb, so a pass that does not
  // accumulate clears those slots first -- one memset when the caller's slots are one array (uncltmo_amd/autograd.py), one per
  // layer otherwise.  UNCL_BWD_FUSED_BIAS=0: separate column-sum kernels as before.
  static const int fused_bias_on = [] { const char* e = getenv("UNCL_BWD_FUSED_BIAS"); return e ? atoi(e) : 1; }();
  c.fused_bias = fused_bias_on && c.dt == UNCL_BF16;
  // (clip layout: every bias slot of the 3x3 / 2x2 layers is ADDED to by the deferred pass, whatever the kernel family)
  if ((c.fused_bias || clip) && !b->accumulate) {
    static const struct { int wi, cout; } k3[] = {{W_INC1, 32}, {W_D0A, 64}, {W_D0B, 64}, {W_D1A, 128}, {W_D1B, 128}, {W_D2A, 256},
                                                  {W_D2B, 256}, {W_D3A, 256}, {W_D3B, 256}, {W_U0A, 128}, {W_U0B, 128}, {W_U1A, 64},
                                                  {W_U1B, 64}, {W_U2A, 32}, {W_U2B, 32}, {W_U3A, 32}, {W_U3B, 32}};
    float* lo = b->gb[W_INC1];
    float* hi = b->gb[W_U3B] + 32;
    // ONE array in weight order, every slot exactly behind its predecessor (uncltmo_amd/autograd.py lays them out so)?  Then one
    // memset covers them -- the graph block's and the up-convs' slots in between belong to the same array and are overwritten
    // later in the pass.  Anything else (separately allocated slots, however close) is cleared slot by slot: the bytes between
    // two slots are not this library's to zero.
    static const int kBias[UNCL_G_NUM_WEIGHTS] = {32, 64, 64, 128, 128, 256, 256, 256, 256, 256, 512, 256, 256, 256, 256, 128, 128,
                                                  128, 64, 64, 64, 32, 32, 32, 32, 32};
    bool contiguous = true;
    for (int i = 0; contiguous && i + 1 < UNCL_G_NUM_WEIGHTS; ++i) contiguous = b->gb[i + 1] == b->gb[i] + kBias[i];
    if (contiguous) {
      if (hipMemsetAsync(lo, 0, (size_t)(hi - lo) * sizeof(float), c.s) != hipSuccess) return UNCL_ERR_LAUNCH;
    } else if (clip) {
      for (int i = 0; i < UNCL_G_NUM_WEIGHTS; ++i)
        if (hipMemsetAsync(b->gb[i], 0, (size_t)kBias[i] * sizeof(float), c.s) != hipSuccess) return UNCL_ERR_LAUNCH;
    } else {
      for (const auto& e : k3)
        if (hipMemsetAsync(b->gb[e.wi], 0, (size_t)e.cout * sizeof(float), c.s) != hipSuccess) return UNCL_ERR_LAUNCH;
    }
  }
  int rc = backward_all(c);
  if (rc == UNCL_OK && clip && b->clip_t == 0) {
    const int rcf = c.flush_colsums();        // this frame's staged bias sums (the queue is per call)
    if (rcf != UNCL_OK) return rcf;
    BCtx ca = c;
    ca.defer = false;
    ca.n = NA;
    ca.L = Lall; ca.Lp = Lall;
    ca.pws = nullptr;
    rc = deferred_wgrads(ca, b->N, b->clip_T);
  }
  const int rc2 = c.flush_colsums();          // joins the weight-gradient stream
  return rc != UNCL_OK ? rc : rc2;
}
