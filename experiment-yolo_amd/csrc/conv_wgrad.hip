// Weight gradient of the convolution on MFMA: dW[tap][co][ci] = sum_pixels dY[pix][co] * X[pix + tap][ci].
//
// Replaces the weight half of ATen convolution_backward for Conv (reference nn/modules/conv.py:41-55).  The GEMM
// reduces over PIXELS, so both MFMA operands need 8 consecutive pixels of one channel per lane while the tensors are
// pixel-major (NHWC): fragments come from LDS through ds_read_b64_tr_b16 (hardware transpose), two reads per
// fragment.  Workgroups are persistent over pixel tiles and keep dW in accumulators; each writes ONE fp32 slab,
// reduced (deterministically, no atomics) by dy_wgrad_reduce into the OIHW fp32 gradient.
#include <cstdio>
#include "common.h"
#include "dealyolo_hip.h"

typedef short short4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct WgArgs {
  const f16* x;
  const f16* dy;
  float* slabs;
  int ldx, lddy;
  int N, H, W, Ho, Wo;
  int cin_p, cout_p;  // padded to multiples of 16
  int cin_r8, cout_r8;  // physical channel counts of x / dy (multiples of 8); granules beyond are read as zero
  int nci_chunks;     // blockIdx.y = co_chunk * nci_chunks + ci_chunk
  int tiles_x, tiles_y, ntiles;
  long npix;  // FLAT only
  // BNF kernels: dy is the gradient w.r.t. the ACTIVATED output; the kernel forms d(raw conv output) itself while staging
  // (BatchNorm + activation backward, bn_act.hip's apply pass) and writes it to draw for the input-gradient pass that follows
  const f16* raw;       // (N,Ho,Wo,ldraw) raw conv output of the forward pass
  f16* draw;            // (N,Ho,Wo,ldraw) out: d(raw)
  const float* coef;    // [4][cout]: scale, shift, mean, invstd
  const double* acc;    // [DY_BN_COPIES][2][cout] sums of g and g*xhat (dy_bn_act_bwd_reduce_acc)
  float* dgamma;
  float* dbeta;
  int ldraw, cout;
  float count;
  // BNF 3 (1x1 only): x is a never-materialised concatenation (DySegs, include/dealyolo_hip.h): every staged granule comes from the
  // segment that holds its channels, through the thread's own pointer (a workgroup's Cin chunk may straddle segments)
  DySegs xs;
  // BNF 1 / 3, 1x1 only: dy lives in TWO planes of one allocation (C2f.cv1's chunk halves: bn_act.hip, ApplyArgs): channels
  // [dy_csplit, cout) start dy_plane bytes behind dy, both planes have pixel stride lddy.  0 = one tensor.  (Pixels past the end of
  // plane 0 would read plane 1 -- the BN forms zero such granules by the RAW tensor's range check, which is why only they take planes.)
  int dy_csplit;
  unsigned dy_plane;
};

static __device__ __forceinline__ half8 tr_frag(const char* base0, const char* base1) {
  // two 4-row x 16-column transposed reads -> 8 consecutive k (pixels) of this lane's channel
  union {
    short4v s[2];
    half8 h;
  } u;
  u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS short4v*)(base0));
  u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS short4v*)(base1));
  return u.h;
}

// output rows per tile: small channel counts get taller tiles so that staging is amortised over more k-steps
static constexpr __host__ __device__ int wg_th(int ks, int stride, int nci, int mtc) {
  if (ks == 1) return (nci + mtc <= 4) ? 8 : 4;
  if (stride == 1) return (nci * mtc <= 1) ? 16 : ((nci + mtc <= 4) ? 8 : 4);
  return (nci * mtc <= 1) ? 8 : ((nci + mtc <= 4) ? 4 : 2);
}

template <int KS, int STRIDE, int NCI, int MTC, int BNF>
// (256, 2): two workgroups per CU.  Without the second argument hipcc budgets 512 registers per lane (it put the 36 accumulator
// tiles of the 64x64 3x3 case into AGPRs: 234 + 188), which leaves ONE workgroup per CU whose four waves stage, wait and
// multiply in lock-step -- 29 % MFMA utilisation.  With two resident workgroups one stages while the other multiplies.
// (The 64x64-channel 3x3 case needs 144 accumulator registers and cannot: it keeps one workgroup per CU; splitting its input
// channels over two workgroups to fit was measured slower, 258 vs 224 us.)
__global__ __launch_bounds__(256, (KS == 3 && NCI * MTC >= 16) ? 1 : 2) void conv_wgrad_kernel(const WgArgs a) {
  constexpr bool FLAT = (KS == 1);
  constexpr bool BN1 = (BNF == 1 || BNF == 3);   // BatchNorm + SiLU backward formed while staging dY
  constexpr bool XSEG = (BNF == 3);              // ... and X is a segmented concatenation
  constexpr int TAPS = KS * KS;
  constexpr int NCOL = NCI * TAPS;          // (ci tile, tap) columns of this workgroup
  constexpr int CPW = (NCOL + 3) / 4;       // columns per wave (round-robin)
  constexpr int TH = wg_th(KS, STRIDE, NCI, MTC), TW = 32;
  constexpr int HWX = FLAT ? TH * TW : (TW - 1) * STRIDE + KS;
  constexpr int HHX = FLAT ? 1 : (TH - 1) * STRIDE + KS;
  constexpr int PAD = KS / 2;
  constexpr int CIN_C = 16 * NCI, COUT_C = 16 * MTC;
  // LDS pitch per staged pixel.  A transposed read touches 8 consecutive pixel rows x 32 bytes per 32-lane group (see load_a),
  // which tile the 64 banks exactly when (row distance in dwords) mod 64 is an odd multiple of 8: pitch = data + 32 B for
  // unit-stride rows, data + 16 B where consecutive k are two pixels apart (stride-2 X tile).  The previous layout (rows
  // q*8+qq, pitch data + 16) measured SQ_LDS_BANK_CONFLICT = 42 % of SQ_LDS_IDX_ACTIVE.
  // (in dwords: unit-stride pitch = 8 mod 16 -> 16/32/48/64 channels take 8/24/24/40; stride-2 pitch = 4 mod 8 -> data + 4)
  constexpr int PSX = (STRIDE == 2) ? CIN_C * 2 + 16 : (((CIN_C / 2 + 7) / 16) * 16 + 8) * 4;
  constexpr int PSY = (((COUT_C / 2 + 7) / 16) * 16 + 8) * 4;
  constexpr int XBYTES = HHX * HWX * PSX, YBYTES = TH * TW * PSY;
  __shared__ __attribute__((aligned(16))) char smem[XBYTES + YBYTES];
  char* const sx = smem;
  char* const sy = smem + XBYTES;
  __shared__ __attribute__((aligned(16))) float sbn[BN1 ? 4 * COUT_C : (BNF == 2 ? 256 * 8 : 4)];  // BNF 1: [sc | sh | kb | kc] of this cout chunk; 2: bias-sum scratch
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // BNF 2: sum over pixels of this thread's dY granule (its channel part is fixed)
  (void)bsum;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  const int qq = p >> 2, pp = p & 3;  // address-supplier role inside the 16-lane group
  const int co_chunk = blockIdx.y / a.nci_chunks, ci_chunk = blockIdx.y - co_chunk * a.nci_chunks;
  const int ci0 = ci_chunk * CIN_C, co0 = co_chunk * COUT_C;

  f32x4 acc[MTC][CPW];
#pragma unroll
  for (int m = 0; m < MTC; ++m)
#pragma unroll
    for (int c = 0; c < CPW; ++c) acc[m][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- software pipeline: the next tile's X halo and dY tile are fetched into registers while the current tile is
  // multiplied (one workgroup keeps ~30-60 KB of loads in flight under its MFMAs instead of idling on them)
  constexpr int NGX = HHX * HWX * (CIN_C / 8), NGY = TH * TW * (COUT_C / 8);
  constexpr int NPX = (NGX + 255) / 256, NPY = (NGY + 255) / 256;
  uint4 pfx[NPX], pfy[NPY];
  uint4 pfr[BN1 ? NPY : 1];     // BNF: raw conv output granules beside the dy granules
  unsigned roff[BN1 ? NPY : 1]; // their byte offsets in the (raw / draw) geometry
  int ncur = 0, nnext = 0;  // BNF, 3x3: image index of the staged / prefetched tile (the draw store needs its descriptor)
  (void)ncur; (void)nnext;
  unsigned vnext = 0, vcur = 0, ornext = 0, orcur = 0;  // BNF: validity bits + (raw / draw) tile-origin offset of the prefetched / staged tile
  (void)pfr; (void)roff; (void)vnext; (void)vcur; (void)ornext; (void)orcur;
  // tile-invariant byte offset of each of this thread's granules from the tile origin (NEVER = granule does not exist:
  // the buffer range check answers it with zeros, as it does for rows above/below the image and pixels past the end)
  constexpr unsigned NEVER = 0x80000000u;
  constexpr int CPGX = CIN_C / 8, CPGY = COUT_C / 8;
  unsigned xoff[NPX], yoff[NPY];
#pragma unroll
  for (int i = 0; i < NPX; ++i) {
    const int id = tid + i * 256;
    const int pixel = id / CPGX, part = id - pixel * CPGX;
    const bool ok = id < NGX && ci0 + part * 8 < a.cin_r8;
    const int hy = FLAT ? 0 : pixel / HWX, hx = FLAT ? pixel : pixel - hy * HWX;
    xoff[i] = ok ? (unsigned)(((hy * a.W + hx) * a.ldx + ci0 + part * 8) * 2) : NEVER;
  }
  const char* xb[XSEG ? NPX : 1];  // XSEG: where this thread's granule i lives (its segment's base at the granule's channel) ...
  int xl[XSEG ? NPX : 1];          // ... and that segment's pixel stride in bytes
  unsigned xup = 0;                // ... bit i: granule i's segment is a 2x nearest up-sampling of a (N, H/2, W/2) tensor (DySegs.acc = 2)
  bool any_up = false;             // (kernel-uniform) some segment is
  if (XSEG)
    for (int k = 0; k < a.xs.nseg; ++k) any_up = any_up || (a.xs.acc[k] & 2);
  (void)xb; (void)xl; (void)xup; (void)any_up;
  if (XSEG) {
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int id = tid + i * 256;
      const int part = id % CPGX, ch = ci0 + part * 8;
      int sg = 0;
      while (sg + 1 < a.xs.nseg && ch >= a.xs.c_end[sg]) ++sg;
      const int cb = sg ? a.xs.c_end[sg - 1] : 0;
      const bool ok = id < NGX && ch < a.cin_r8;
      xb[XSEG ? i : 0] = ok ? reinterpret_cast<const char*>(a.xs.ptr[sg]) + (ch - cb) * 2 : nullptr;
      xl[XSEG ? i : 0] = a.xs.ld[sg] * 2;
      if (ok && (a.xs.acc[sg] & 2)) xup |= 1u << i;
    }
  }
#pragma unroll
  for (int i = 0; i < NPY; ++i) {
    const int id = tid + i * 256;
    const int pixel = id / CPGY, part = id - pixel * CPGY;
    const bool ok = id < NGY && co0 + part * 8 < a.cout_r8;
    const int ty = FLAT ? 0 : pixel / TW, tx = FLAT ? pixel : pixel - ty * TW;
    yoff[i] = ok ? (unsigned)(((ty * a.Wo + tx) * a.lddy + co0 + part * 8) * 2) : NEVER;
    if (FLAT && BN1 && a.dy_csplit && ok) {
      const int ch = co0 + part * 8;
      yoff[i] = (unsigned)((tx * a.lddy + (ch < a.dy_csplit ? ch : ch - a.dy_csplit)) * 2) + (ch < a.dy_csplit ? 0u : a.dy_plane);
    }
    if (BN1) roff[i] = ok ? (unsigned)(((ty * a.Wo + tx) * a.ldraw + co0 + part * 8) * 2) : NEVER;
  }
  if (BN1) {
    // BatchNorm backward coefficients of this cout chunk from the reduce pass's fp64 sums (what bn_act_bwd_apply_kernel<.., true>
    // does in its prologue): dx = sc*g - (kb*x + kc), kb = sc*invstd*mean(g*xhat), kc = sc*mean(g) - kb*mean
    for (int c = tid; c < COUT_C; c += 256) {
      const int ch = co0 + c;
      float sc = 0.f, sh = 0.f, kb = 0.f, kc = 0.f;
      if (ch < a.cout) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < DY_BN_COPIES; ++k) {
          s1 += a.acc[(size_t)(k * 2 + 0) * a.cout + ch];
          s2 += a.acc[(size_t)(k * 2 + 1) * a.cout + ch];
        }
        const float mg = (float)(s1 / a.count), mgx = (float)(s2 / a.count);
        sc = a.coef[ch];
        sh = a.coef[a.cout + ch];
        const float mean = a.coef[2 * a.cout + ch], inv = a.coef[3 * a.cout + ch];
        kb = sc * inv * mgx;
        kc = sc * mg - kb * mean;
        if (blockIdx.x == 0 && ci_chunk == 0) {
          if (a.dbeta) a.dbeta[ch] = (float)s1;
          if (a.dgamma) a.dgamma[ch] = (float)s2;
        }
      }
      sbn[c] = sc;
      sbn[COUT_C + c] = sh;
      sbn[2 * COUT_C + c] = kb;
      sbn[3 * COUT_C + c] = kc;
    }
    __syncthreads();
  }
  auto ld16 = [](__amdgpu_buffer_rsrc_t r, unsigned off) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return *reinterpret_cast<const uint4*>(&v);
  };
  auto prefetch = [&](int tile) {
    if (FLAT) {
      const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(a.x), 0, (int)((unsigned)a.npix * (unsigned)a.ldx * 2u), 0x00020000);
      const auto ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(a.dy), 0, (int)((unsigned)a.npix * (unsigned)a.lddy * 2u + (a.dy_csplit ? a.dy_plane : 0u)), 0x00020000);
      const unsigned ox = (unsigned)tile * (TH * TW) * a.ldx * 2u, oy = (unsigned)tile * (TH * TW) * a.lddy * 2u;
      if (XSEG) {
        const long p0 = (long)tile * (TH * TW);
        // up-sampled members: pixel (n, y, x) of the concatenation reads (n, y >> 1, x >> 1) of the low-resolution tensor.  The tile's
        // first pixel is decomposed once (uniform integer divisions), a lane's pixel lies < TH * TW further: exact float divisions
        int n0 = 0, y0 = 0, x0 = 0;
        if (any_up) {
          const int hw = a.H * a.W, g0 = (int)p0;
          n0 = g0 / hw;
          const int r0 = g0 - n0 * hw;
          y0 = r0 / a.W;
          x0 = r0 - y0 * a.W;
        }
        const float invW = 1.0f / (float)a.W, invH = 1.0f / (float)a.H;
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
          const int lp = (tid + i * 256) / CPGX;
          long gp = p0 + lp;
          const bool inside = gp < a.npix;
          if ((xup >> i) & 1u) {
            const int xx = x0 + lp, dy_ = (int)(((float)xx + 0.5f) * invW), x = xx - dy_ * a.W;
            const int yt = y0 + dy_, dn = (int)(((float)yt + 0.5f) * invH), y = yt - dn * a.H;
            gp = ((long)(n0 + dn) * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1);
          }
          const char* src = xb[XSEG ? i : 0];
          pfx[i] = (src && inside) ? *reinterpret_cast<const uint4*>(src + gp * xl[XSEG ? i : 0]) : make_uint4(0, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NPX; ++i) pfx[i] = ld16(rx, xoff[i] + ox);
      }
#pragma unroll
      for (int i = 0; i < NPY; ++i) pfy[i] = ld16(ry, yoff[i] + oy);
      if (BN1) {
        const unsigned nbytes = (unsigned)a.npix * (unsigned)a.ldraw * 2u;
        const auto rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(a.raw), 0, (int)nbytes, 0x00020000);
        ornext = (unsigned)tile * (TH * TW) * a.ldraw * 2u;
        vnext = 0;
#pragma unroll
        for (int i = 0; i < NPY; ++i) {
          const unsigned off = roff[i] + ornext;
          pfr[i] = ld16(rr, off);
          vnext |= (roff[i] != NEVER && off < nbytes) ? (1u << i) : 0u;
        }
      }
    } else {
      const int bx = tile % a.tiles_x;
      const int t2 = tile / a.tiles_x;
      const int n = t2 / a.tiles_y;
      const int oy0 = (t2 % a.tiles_y) * TH, ox0 = bx * TW;
      const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;
      const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(a.x) + (size_t)n * a.H * a.W * a.ldx, 0, a.H * a.W * a.ldx * 2, 0x00020000);
      const auto ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(a.dy) + (size_t)n * a.Ho * a.Wo * a.lddy, 0, a.Ho * a.Wo * a.lddy * 2, 0x00020000);
      const unsigned ox = (unsigned)((iy0 * a.W + ix0) * a.ldx * 2), oy = (unsigned)((oy0 * a.Wo + ox0) * a.lddy * 2);
      if (ix0 >= 0 && ix0 + HWX <= a.W) {  // wave-uniform: interior in x, no column masks
#pragma unroll
        for (int i = 0; i < NPX; ++i) pfx[i] = ld16(rx, xoff[i] + ox);
      } else {
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
          const int hx = ((tid + i * 256) / CPGX) % HWX;  // recomputed on border tiles only
          pfx[i] = ld16(rx, (unsigned)(ix0 + hx) < (unsigned)a.W ? xoff[i] + ox : NEVER);
        }
      }
      if (ox0 + TW <= a.Wo) {
#pragma unroll
        for (int i = 0; i < NPY; ++i) pfy[i] = ld16(ry, yoff[i] + oy);
      } else {
#pragma unroll
        for (int i = 0; i < NPY; ++i) {
          const int tx = ((tid + i * 256) / CPGY) % TW;
          pfy[i] = ld16(ry, ox0 + tx < a.Wo ? yoff[i] + oy : NEVER);
        }
      }
      if (BN1) {
        const unsigned nbytes = (unsigned)(a.Ho * a.Wo * a.ldraw * 2);
        const auto rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(a.raw) + (size_t)n * a.Ho * a.Wo * a.ldraw, 0, (int)nbytes, 0x00020000);
        ornext = (unsigned)((oy0 * a.Wo + ox0) * a.ldraw * 2);
        vnext = 0;
        const bool xfull = ox0 + TW <= a.Wo;
#pragma unroll
        for (int i = 0; i < NPY; ++i) {
          unsigned off = roff[i] == NEVER ? NEVER : roff[i] + ornext;
          if (!xfull) {
            const int tx = ((tid + i * 256) / CPGY) % TW;
            off = ox0 + tx < a.Wo ? off : NEVER;
          }
          pfr[i] = ld16(rr, off);
          vnext |= off < nbytes ? (1u << i) : 0u;
        }
        nnext = n;
      }
    }
  };


  // Workgroups are dealt round-robin to the 8 XCDs (one L2 each): with tile = blockIdx.x + k * gridDim.x the vertical neighbours of
  // a 3x3 tile sit on other XCDs and every halo row of X is fetched into two L2s.  Giving XCD x the x-th eighth of each period's
  // tiles keeps them together (the map of conv.hip's ping-pong kernel).
  int tile = (!FLAT && !(gridDim.x & 7)) ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile < a.ntiles) prefetch(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();  // previous tile's fragment reads are done
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int id = tid + i * 256;
      if (id < NGX) *reinterpret_cast<uint4*>(sx + (id / (CIN_C / 8)) * PSX + (id % (CIN_C / 8)) * 16) = pfx[i];
    }
    if (BN1) {
      vcur = vnext; orcur = ornext; ncur = nnext;
      // d(raw) = BatchNorm + SiLU backward of (dy, raw), formed here in registers: it becomes the A operand in LDS and -- from the
      // workgroups of Cin chunk 0 -- the tensor the input-gradient pass reads.  Granules the range check zeroed (outside the image,
      // past the last pixel, absent channels) must stay zero: with dy = raw = 0 the formula gives -kc, not 0.
      const unsigned nbytes = FLAT ? (unsigned)a.npix * (unsigned)a.ldraw * 2u : (unsigned)(a.Ho * a.Wo * a.ldraw * 2);
      const auto rd = __builtin_amdgcn_make_buffer_rsrc(FLAT ? a.draw : a.draw + (size_t)ncur * a.Ho * a.Wo * a.ldraw, 0, (int)nbytes, 0x00020000);
#pragma unroll
      for (int i = 0; i < NPY; ++i) {
        const int id = tid + i * 256;
        const int part = id % CPGY;
        const bool valid = (vcur >> i) & 1u;
        const float* cf = sbn + part * 8;
        union { half8 h; uint4 u; u32x4 v; } o;
        o.h = bn_bwd_apply8<DY_ACT_SILU>(*reinterpret_cast<const half8*>(&pfy[i]), *reinterpret_cast<const half8*>(&pfr[i]), cf,
                                         cf + COUT_C, cf + 2 * COUT_C, cf + 3 * COUT_C);
        if (!valid) o.u = make_uint4(0, 0, 0, 0);
        if (id < NGY) *reinterpret_cast<uint4*>(sy + (id / (COUT_C / 8)) * PSY + (id % (COUT_C / 8)) * 16) = o.u;
        if (ci_chunk == 0 && a.draw) __builtin_amdgcn_raw_buffer_store_b128(o.v, rd, (int)(valid ? roff[i] + orcur : NEVER), 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPY; ++i) {
        const int id = tid + i * 256;
        if (id < NGY) *reinterpret_cast<uint4*>(sy + (id / (COUT_C / 8)) * PSY + (id % (COUT_C / 8)) * 16) = pfy[i];
        if (BNF == 2 && id < NGY) {  // bias gradient = sum of dY over the pixels: taken while the granule passes (zeros outside the map)
          const half8 v = *reinterpret_cast<const half8*>(&pfy[i]);
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[j] += (float)v[j];
        }
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) prefetch(tile + gridDim.x);
    // software-pipelined: A fragments of k-step r+1 and the B fragment of column c+1 are requested before the MFMAs
    // of (r, c) issue, so the transposed LDS reads run under the matrix pipe instead of in front of it
    const int kpix = 4 * q + qq;  // first-read row of this lane inside a k-step; the second read is 16 rows further.  (Which of
                                  // the 32 pixels of a k-step a lane's k index means is free as long as A and B agree: with
                                  // rows 4q+qq one 32-lane group reads 8 CONSECUTIVE rows, which the pitch above makes conflict-free.)
    half8 af[2][MTC], bf[2];
    auto load_a = [&](int buf, int r) {
#pragma unroll
      for (int m = 0; m < MTC; ++m) {
        const char* b0 = sy + (r * TW + kpix) * PSY + (m * 16 + 4 * pp) * 2;
        af[buf][m] = tr_frag(b0, b0 + 16 * PSY);
      }
    };
    auto load_b = [&](int buf, int r, int c) {
      const int col = wave + 4 * c;
      if (col < NCOL) {
        const int cit = col / TAPS, tap = col - cit * TAPS;
        const int dy = tap / KS, dx = tap - dy * KS;
        const char* b0;
        int step;
        if (FLAT) {
          b0 = sx + (r * TW + kpix) * PSX + (cit * 16 + 4 * pp) * 2;
          step = 16 * PSX;
        } else {
          b0 = sx + ((r * STRIDE + dy) * HWX + kpix * STRIDE + dx) * PSX + (cit * 16 + 4 * pp) * 2;
          step = 16 * STRIDE * PSX;
        }
        bf[buf] = tr_frag(b0, b0 + step);
      }
    };
    load_a(0, 0);
    load_b(0, 0, 0);
#pragma unroll
    for (int r = 0; r < TH; ++r) {  // one k-step = 32 consecutive output pixels of tile row r
      if (r + 1 < TH) load_a((r + 1) & 1, r + 1);
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const int lin = r * CPW + c;
        if (c + 1 < CPW) load_b((lin + 1) & 1, r, c + 1);
        else if (r + 1 < TH) load_b((lin + 1) & 1, r + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (wave + 4 * c < NCOL) {
#pragma unroll
          for (int m = 0; m < MTC; ++m)
            acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[r & 1][m], bf[lin & 1], acc[m][c], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  if (BNF == 2 && ci_chunk == 0) {  // (256 % CPGY == 0 is the launcher's condition: a thread's channel part never changes)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) sbn[tid * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < COUT_C && co0 + tid < a.cout) {
      const int part = tid >> 3, j = tid & 7;
      float t = 0.f;
      for (int k = part; k < 256; k += CPGY) t += sbn[k * 8 + j];
      unsafeAtomicAdd(const_cast<double*>(a.acc) + (size_t)(blockIdx.x % DY_BN_COPIES) * a.cout + co0 + tid, (double)t);
    }
  }
  // ---- one slab per workgroup: [tap][cout_p][cin_p] fp32 (16 lanes -> 64 contiguous bytes)
  float* slab = a.slabs + (size_t)blockIdx.x * TAPS * a.cout_p * a.cin_p;
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int col = wave + 4 * c;
    if (col < NCOL) {
      const int cit = col / TAPS, tap = col - cit * TAPS;
#pragma unroll
      for (int m = 0; m < MTC; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + m * 16 + q * 4 + r, ci = ci0 + cit * 16 + p;
          slab[((size_t)tap * a.cout_p + co) * a.cin_p + ci] = acc[m][c][r];
        }
    }
  }
}

// dW[co][ci][tap] (+)= sum_wg slab[wg][tap][co][ci].  One wave covers 64 consecutive slab elements (coalesced 256-byte
// reads per slab); the slab axis is split over the block's 4 waves with 4 independent loads in flight per thread, so
// the reduction is bandwidth- rather than latency-bound.
static __device__ __forceinline__ void wgrad_reduce_block(int block, const float* slabs, int nslabs, float* dw, int cout, int cin,
                                                          int taps, int cout_p, int cin_p, int accumulate,
                                                          int ld_taps, int ld_cphys, int ld_cin) {
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long S = (long)taps * cout_p * cin_p;
  const long o = (long)block * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (o < S) {
    const float* p = slabs + o;
    int k = grp;
    for (; k + 12 < nslabs; k += 16) {
      s0 += p[(long)k * S];
      s1 += p[(long)(k + 4) * S];
      s2 += p[(long)(k + 8) * S];
      s3 += p[(long)(k + 12) * S];
    }
    for (; k < nslabs; k += 4) s0 += p[(long)k * S];
  }
  __shared__ float red[4][64];
  red[grp][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && o < S) {
    const float tot = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int ci = (int)(o % cin_p);
    const long t = o / cin_p;
    const int co = (int)(t % cout_p), tap = (int)(t / cout_p);
    if (ld_taps) {  // LDConv column conv: ci = n*ld_cphys + channel  ->  dw[(co*ld_cin + channel)*ld_taps + n]
      const int n = ci / ld_cphys, ch = ci - n * ld_cphys;
      if (co < cout && n < ld_taps && ch < ld_cin) {
        float* d = dw + ((long)co * ld_cin + ch) * ld_taps + n;
        *d = accumulate ? *d + tot : tot;
      }
    } else if (co < cout && ci < cin) {
      float* d = dw + ((long)co * cin + ci) * taps + tap;
      *d = accumulate ? *d + tot : tot;
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* slabs, int nslabs, float* dw, int cout, int cin,
                                                           int taps, int cout_p, int cin_p, int accumulate,
                                                           int ld_taps, int ld_cphys, int ld_cin) {
  wgrad_reduce_block(blockIdx.x, slabs, nslabs, dw, cout, cin, taps, cout_p, cin_p, accumulate, ld_taps, ld_cphys, ld_cin);
}

// Every layer's slab reduction of one backward pass in ONE launch (67 launches of ~10 us, each latency-bound on a few
// hundred KB, become one bandwidth-bound pass).  descs[i].first_block is the exclusive prefix of per-layer block counts.
struct WgReduceDesc {
  const float* slabs;
  float* dw;
  int nslabs, cout, cin, taps, cout_p, cin_p, accumulate, ld_taps, ld_cphys, ld_cin;
  int first_block, bias_c;      // bias_c > 0: dbias[c] = sum over the DY_BN_COPIES copies of bias_acc (dy_conv_wgrad_bias)
  const double* bias_acc;
  float* dbias;
};
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const WgReduceDesc* descs, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {  // last descriptor whose first_block <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].first_block <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const WgReduceDesc d = descs[lo];
  if (d.bias_c > 0 && (int)blockIdx.x == d.first_block && (int)threadIdx.x < d.bias_c) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < DY_BN_COPIES; ++k) t += d.bias_acc[(size_t)k * d.bias_c + threadIdx.x];
    d.dbias[threadIdx.x] = (float)t;
  }
  wgrad_reduce_block(blockIdx.x - d.first_block, d.slabs, d.nslabs, d.dw, d.cout, d.cin, d.taps, d.cout_p, d.cin_p, d.accumulate,
                     d.ld_taps, d.ld_cphys, d.ld_cin);
}
extern "C" int dy_wgrad_reduce_desc_bytes(void) { return (int)sizeof(WgReduceDesc); }
static void wgrad_geometry(int cin, int cout, int ks, int stride, int* cin_p, int* cout_p, int* nci, int* mtc, long npix_out = 0);
// Fills one host-side descriptor for a layer whose dy_conv_wgrad / dy_conv_wgrad_ld call was made with dw == NULL; returns its
// block count (to be prefix-summed into first_block by the caller) or a negative error.
extern "C" int dy_wgrad_reduce_desc_fill(void* desc, const float* slabs, int nslabs, float* dw, int cin, int cout, int ks, int stride,
                                         int accumulate, int ld_taps, int ld_cphys, int ld_cin, int first_block) {
  if (!desc || !slabs || !dw || nslabs < 1) return DY_ERR_ARG;
  int cp, op, nci, mtc;
  const int cin_eff = ld_taps ? ld_taps * ld_cphys : cin;
  wgrad_geometry(cin_eff, cout, ld_taps ? 1 : ks, ld_taps ? 1 : stride, &cp, &op, &nci, &mtc);
  const int taps = ld_taps ? 1 : ks * ks;
  WgReduceDesc* d = reinterpret_cast<WgReduceDesc*>(desc);
  *d = WgReduceDesc{slabs, dw, nslabs, cout, cin_eff, taps, op, cp, accumulate, ld_taps, ld_cphys, ld_cin, first_block, 0, nullptr, nullptr};
  return cdiv(taps * op * cp, 64);
}
// a descriptor whose layer ran dy_conv_wgrad_bias: the same reduction launch also finishes that layer's bias gradient
extern "C" int dy_wgrad_reduce_desc_bias(void* desc, const double* bias_acc, float* dbias, int c) {
  if (!desc || !bias_acc || !dbias || c < 1 || c > 256) return DY_ERR_ARG;
  WgReduceDesc* d = reinterpret_cast<WgReduceDesc*>(desc);
  d->bias_acc = bias_acc; d->dbias = dbias; d->bias_c = c;
  return DY_OK;
}
extern "C" int dy_wgrad_reduce_batched(const void* descs_device, int n, int total_blocks, hipStream_t stream) {
  if (n <= 0 || total_blocks <= 0 || !descs_device) return DY_ERR_ARG;
  hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(total_blocks), dim3(256), 0, stream, (const WgReduceDesc*)descs_device, n);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

template <int KS, int STRIDE, int NCI, int MTC>
static int launch_wgrad(const WgArgs& a, int gx, int gy, hipStream_t s) {
  if (a.raw && a.xs.nseg > 0) {
    if constexpr (KS == 1) hipLaunchKernelGGL((conv_wgrad_kernel<KS, STRIDE, NCI, MTC, 3>), dim3(gx, gy), dim3(256), 0, s, a);
    else return DY_ERR_ARG;
  } else if (a.raw) hipLaunchKernelGGL((conv_wgrad_kernel<KS, STRIDE, NCI, MTC, 1>), dim3(gx, gy), dim3(256), 0, s, a);
  else if (a.acc && MTC != 3) hipLaunchKernelGGL((conv_wgrad_kernel<KS, STRIDE, NCI, MTC, 2>), dim3(gx, gy), dim3(256), 0, s, a);
  else if (a.acc) return DY_ERR_ARG;
  else hipLaunchKernelGGL((conv_wgrad_kernel<KS, STRIDE, NCI, MTC, 0>), dim3(gx, gy), dim3(256), 0, s, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
template <int KS, int STRIDE>
static int dispatch_wgrad(int nci, int mtc, const WgArgs& a, int gx, int gy, hipStream_t s) {
#define DY_CASE(I, M) \
  if (nci == I && mtc == M) return launch_wgrad<KS, STRIDE, I, M>(a, gx, gy, s);
  DY_CASE(1, 1) DY_CASE(1, 2) DY_CASE(1, 4)
  DY_CASE(2, 1) DY_CASE(2, 2) DY_CASE(2, 4)
  if (!(KS == 3 && STRIDE == 2)) { DY_CASE(4, 1) DY_CASE(4, 2) DY_CASE(4, 4) }
  if (KS == 1) {  // 48- / 96-channel sides of the C2f 1x1 convs: three 16-channel tiles in one workgroup instead of three passes
    DY_CASE(3, 1) DY_CASE(3, 2) DY_CASE(3, 4) DY_CASE(1, 3) DY_CASE(2, 3) DY_CASE(4, 3) DY_CASE(3, 3)
  }
#undef DY_CASE
  return DY_ERR_ARG;
}

// (ci tiles, co tiles) of 16 channels one workgroup owns.  Every resident workgroup column ends by writing a full fp32 weight slab
// that the reduce launch reads back, so the slab bytes of a layer are (workgroup columns) x (weight bytes) whatever the map size:
// 37.7 MB for a 64->64 3x3, against 26 MB of activations on a 40x40 map at batch 64.  npix_out > 0 (the layer's output pixels) lets
// small maps trade that for re-reads of the activations, which stay in L2 / Infinity Cache at that size: a workgroup then owns a
// smaller (ci, co) block over more pixels, fewer columns exist, and every operand is read once per chunk of the OTHER side.
// Halves the co side first (a second ci chunk repeats the BatchNorm backward arithmetic of dy_conv_wgrad_bn, a second co chunk only
// re-reads x).  DY_WGRAD_SPLIT=0 restores one block per layer; DY_WGRAD_SPLIT_F scales the threshold: measured per step (one box,
// alternating runs) off 12.810 / 12.740 ms, F=0.5 13.330 / 13.217 (re-reads dominate), F=1 12.787 / 12.745, F=2 12.708 / 12.676
// (the default: 64->64 3x3 @40x40 takes (32, 32) blocks, 128 columns), F=4 12.778 / 12.701.
// resident workgroups per CU the grid is sized for: 1 where the accumulators leave room for one (64x64 3x3), 2 otherwise -- and, for the
// lightest blocks (one or two 16x16 tiles: the 16-channel Bottlenecks at 160x160), DY_WGRAD_LIGHT_WGS (their slabs are a few KB, their
// LDS tiles ~36 KB and their registers < 128, so more of them fit and hide each other's staging latency)
static int wgrad_per_cu(int ks, int nci, int mtc) {
  static const int light = getenv("DY_WGRAD_LIGHT_WGS") ? atoi(getenv("DY_WGRAD_LIGHT_WGS")) : 2;
  if (ks == 3 && nci * mtc >= 16) return 1;
  return nci * mtc <= 2 ? light : 2;
}
static long wgrad_columns(int ks, int nci, int mtc, int cp, int op) {
  const int gy = (cp / (16 * nci)) * (op / (16 * mtc));
  const int per_cu = wgrad_per_cu(ks, nci, mtc);
  int gx = (256 * per_cu) / gy;
  return gx < 32 ? 32 : gx;
}
static void wgrad_geometry(int cin, int cout, int ks, int stride, int* cin_p, int* cout_p, int* nci, int* mtc, long npix_out) {
  *cin_p = (cin + 15) / 16 * 16;
  *cout_p = (cout + 15) / 16 * 16;
  const int cit = *cin_p / 16, cot = *cout_p / 16;
  const int cap = (ks == 3 && stride == 2) ? 2 : 4;
  *nci = (cit % 4 == 0 && cap >= 4) ? 4 : ((ks == 1 && cit % 3 == 0) ? 3 : (cit % 2 == 0 ? 2 : 1));
  *mtc = cot % 4 == 0 ? 4 : ((ks == 1 && cot % 3 == 0) ? 3 : (cot % 2 == 0 ? 2 : 1));
  static const bool split = !(getenv("DY_WGRAD_SPLIT") && atoi(getenv("DY_WGRAD_SPLIT")) == 0);
  static const double f = getenv("DY_WGRAD_SPLIT_F") ? atof(getenv("DY_WGRAD_SPLIT_F")) : 2.0;
  if (!split || npix_out <= 0) return;
  const double wbytes = (double)ks * ks * *cin_p * *cout_p * 4.0;
  const double act = (double)npix_out * (*cin_p * stride * stride + *cout_p) * 2.0;
  while (2.0 * wgrad_columns(ks, *nci, *mtc, *cin_p, *cout_p) * wbytes > f * act) {  // slab written + read back
    if (*mtc % 2 == 0 && *mtc >= *nci) *mtc /= 2;
    else if (*nci % 2 == 0) *nci /= 2;
    else if (*mtc % 2 == 0) *mtc /= 2;
    else break;
  }
}

extern "C" int dy_wgrad_workspace(int n, int h, int w, int cin, int cout, int ks, int stride, int* nslabs,
                                  long* slab_elems) {
  int cp, op, nci, mtc;
  const int pad = ks / 2, Ho = (h + 2 * pad - ks) / stride + 1, Wo = (w + 2 * pad - ks) / stride + 1;
  wgrad_geometry(cin, cout, ks, stride, &cp, &op, &nci, &mtc, (long)n * Ho * Wo);
  const int th = wg_th(ks, stride, nci, mtc);
  int ntiles;
  if (ks == 1) ntiles = (int)(((long)n * Ho * Wo + th * 32 - 1) / (th * 32));
  else ntiles = cdiv(Wo, 32) * cdiv(Ho, th) * n;
  *slab_elems = (long)ks * ks * cp * op;
  // one slab per workgroup and exactly one resident round of workgroups: 256 CUs x (1 or 2 per CU, see the kernel's launch
  // bounds) over the (ci chunk, co chunk) grid rows.  More workgroups than that only add slab traffic (147 KB each for
  // 64x64 3x3) and a ragged second round; fewer leave CUs idle.
  const int gy = (cp / (16 * nci)) * (op / (16 * mtc));
  const int per_cu = wgrad_per_cu(ks, nci, mtc);
  int gx = (256 * per_cu) / gy;
  if (gx < 32) gx = 32;
  // every workgroup ends by writing a full fp32 weight slab (147 KB for 64x64x3x3) that the reduce kernel reads back: on small
  // maps a workgroup with one or two tiles moves more slab bytes than activation bytes, so keep at least TPB tiles per slab
  static const int tpb = getenv("DY_WGRAD_TPB") ? atoi(getenv("DY_WGRAD_TPB")) : 1;
  if (tpb > 1 && gx > ntiles / tpb) gx = ntiles / tpb > 32 ? ntiles / tpb : (ntiles < 32 ? ntiles : 32);
  if (gx > ntiles) gx = ntiles;
  if (gx < 1) gx = 1;
  *nslabs = gx;
  return DY_OK;
}

struct WgBnHost {  // BatchNorm + SiLU backward folded into the staging of dY (dy_conv_wgrad_bn)
  const void* raw; void* draw; const float* coef; const double* acc; float* dgamma; float* dbeta; int ldraw; float count;
  const DySegs* xs = nullptr;  // dy_conv1x1_wgrad_bn_segs: the X operand is a segmented concatenation
  int dy_csplit = 0;           // dy_conv1x1_wgrad_bn_planes: dY in two planes (WgArgs)
  unsigned dy_plane = 0;
};
static int conv_wgrad_impl(const void* x, int ldx, const void* dy, int lddy, float* slabs, float* dw, int n, int h, int w,
                           int cin, int cout, int ks, int stride, int accumulate, int ld_taps, int ld_cphys, int ld_cin,
                           hipStream_t stream, const WgBnHost* bn = nullptr);

// host-side: the kernel instantiation dy_conv_wgrad launches for this geometry, spelled as rocprofv3 prints it
extern "C" int dy_wgrad_kernel_name(int cin, int cout, int ks, int stride, char* out, int cap) {
  if (!out || cap < 8) return DY_ERR_ARG;
  int cp, op, nci, mtc;
  wgrad_geometry(cin, cout, ks, stride, &cp, &op, &nci, &mtc);
  snprintf(out, cap, "conv_wgrad_kernel<%d, %d, %d, %d, 0>", ks, stride, nci, mtc);  // ", 1>": the dy_conv_wgrad_bn form
  return DY_OK;
}

// the same for a given map (small maps take smaller channel blocks per workgroup: wgrad_geometry)
extern "C" int dy_wgrad_kernel_name_at(int n, int h, int w, int cin, int cout, int ks, int stride, char* out, int cap) {
  if (!out || cap < 8) return DY_ERR_ARG;
  int cp, op, nci, mtc;
  wgrad_geometry(cin, cout, ks, stride, &cp, &op, &nci, &mtc,
                 (long)n * ((h + 2 * (ks / 2) - ks) / stride + 1) * ((w + 2 * (ks / 2) - ks) / stride + 1));
  snprintf(out, cap, "conv_wgrad_kernel<%d, %d, %d, %d, 0>", ks, stride, nci, mtc);
  return DY_OK;
}

extern "C" int dy_conv_wgrad(const void* x, int ldx, const void* dy, int lddy, float* slabs, float* dw, int n, int h,
                             int w, int cin, int cout, int ks, int stride, int accumulate, hipStream_t stream) {
  return conv_wgrad_impl(x, ldx, dy, lddy, slabs, dw, n, h, w, cin, cout, ks, stride, accumulate, 0, 0, 0, stream);
}
// LDConv column conv: x = sampled map with ld_taps*ld_cphys channels, dw = (cout, ld_cin, ld_taps, 1) fp32
extern "C" int dy_conv_wgrad_ld(const void* x, int ldx, const void* dy, int lddy, float* slabs, float* dw, int n, int h,
                                int w, int cout, int ld_cin, int ld_taps, int ld_cphys, int accumulate, hipStream_t stream) {
  return conv_wgrad_impl(x, ldx, dy, lddy, slabs, dw, n, h, w, ld_taps * ld_cphys, cout, 1, 1, accumulate, ld_taps, ld_cphys,
                         ld_cin, stream);
}

// The weight gradient of a Conv (conv + BatchNorm + SiLU, reference nn/modules/conv.py:49-55) given the gradient w.r.t. its
// ACTIVATED output: the BatchNorm / SiLU backward apply pass (dy_bn_act_bwd_apply_acc) runs inside the kernel, on the dY operand
// while it is staged, and d(raw conv output) is written once to `draw` (same geometry as `raw`) for the input-gradient pass.
extern "C" int dy_conv_wgrad_bn(const void* x, int ldx, const void* dy, int lddy, const void* raw, int ldraw, void* draw,
                                const float* coef, const double* acc, float* dgamma, float* dbeta, float count, float* slabs,
                                float* dw, int n, int h, int w, int cin, int cout, int ks, int stride, int accumulate,
                                hipStream_t stream) {
  // draw == NULL: no input-gradient pass follows (the stem), d(raw) stays in the kernel
  if (!raw || !coef || !acc || (ldraw & 7) || ((uintptr_t)raw & 15) || ((uintptr_t)draw & 15) || (cout & 15)) return DY_ERR_ARG;
  const WgBnHost bn{raw, draw, coef, acc, dgamma, dbeta, ldraw, count};
  return conv_wgrad_impl(x, ldx, dy, lddy, slabs, dw, n, h, w, cin, cout, ks, stride, accumulate, 0, 0, 0, stream, &bn);
}
// Weight gradient of a conv WITH BIAS (Detect's final nn.Conv2d, LDConv.p_conv: reference nn/modules/head.py:38-42, conv.py:356):
// the bias gradient -- the sum of dY over the pixels -- is taken from the dY granules while they are staged and added into
// bias_acc [DY_BN_COPIES][cout8] (fp64, zeroed by the caller); dy_wgrad_reduce_batched finishes it (dy_wgrad_reduce_desc_bias).
extern "C" int dy_conv_wgrad_bias(const void* x, int ldx, const void* dy, int lddy, double* bias_acc, float* slabs, float* dw, int n,
                                  int h, int w, int cin, int cout, int ks, int stride, int accumulate, hipStream_t stream) {
  if (!bias_acc) return DY_ERR_ARG;
  const WgBnHost bn{nullptr, nullptr, nullptr, bias_acc, nullptr, nullptr, 0, 1.f};
  return conv_wgrad_impl(x, ldx, dy, lddy, slabs, dw, n, h, w, cin, cout, ks, stride, accumulate, 0, 0, 0, stream, &bn);
}
// dy_conv_wgrad_bn for a 1x1 Conv whose input is the never-materialised concatenation xs (cin = xs->c_end[last]): C2f.cv2 / SPPF.cv2 /
// the Conv behind a Concat layer (reference nn/modules/block.py:222-226, :166-171, nn/modules/conv.py:338-348).
extern "C" int dy_conv1x1_wgrad_bn_segs(const DySegs* xs, const void* dy, int lddy, const void* raw, int ldraw, void* draw, const float* coef,
                                        const double* acc, float* dgamma, float* dbeta, float count, float* slabs, float* dw, int n, int h,
                                        int w, int cin, int cout, int accumulate, hipStream_t stream) {
  if (!raw || !coef || !acc || (ldraw & 7) || ((uintptr_t)raw & 15) || ((uintptr_t)draw & 15) || (cout & 15)) return DY_ERR_ARG;
  if (!xs || xs->nseg < 1 || xs->nseg > DY_MAX_SEGS || xs->c_end[xs->nseg - 1] != cin) return DY_ERR_ARG;
  for (int k = 0; k < xs->nseg; ++k)
    if ((xs->c_end[k] & 7) || (xs->ld[k] & 7) || !xs->ptr[k] || ((uintptr_t)xs->ptr[k] & 15) || xs->c_end[k] <= (k ? xs->c_end[k - 1] : 0) ||
        ((xs->acc[k] & 2) && ((h | w) & 1))) return DY_ERR_ARG;  // (acc = 2: an up-sampled member, read at (y >> 1, x >> 1): DySegs)
  WgBnHost bn{raw, draw, coef, acc, dgamma, dbeta, ldraw, count};
  bn.xs = xs;
  return conv_wgrad_impl(xs->ptr[0], 8, dy, lddy, slabs, dw, n, h, w, cin, cout, 1, 1, accumulate, 0, 0, 0, stream, &bn);
}
// dy_conv_wgrad_bn / dy_conv1x1_wgrad_bn_segs (xs may be NULL: then (x, ldx) is the input) for a 1x1 Conv whose OUTPUT gradient lives in
// two planes of one allocation -- channels [0, csplit) at dy, [csplit, cout) at dy2 > dy, both with pixel stride lddy (C2f.cv1, whose
// halves are tensors of their own: reference nn/modules/block.py:223)
extern "C" int dy_conv1x1_wgrad_bn_planes(const DySegs* xs, const void* x, int ldx, const void* dy, const void* dy2, int lddy, int csplit,
                                          const void* raw, int ldraw, void* draw, const float* coef, const double* acc, float* dgamma,
                                          float* dbeta, float count, float* slabs, float* dw, int n, int h, int w, int cin, int cout,
                                          int accumulate, hipStream_t stream) {
  if (!raw || !coef || !acc || (ldraw & 7) || ((uintptr_t)raw & 15) || ((uintptr_t)draw & 15) || (cout & 15)) return DY_ERR_ARG;
  if (!dy || !dy2 || (csplit & 7) || csplit <= 0 || csplit >= cout || ((uintptr_t)dy2 & 15)) return DY_ERR_ARG;
  const long plane = (const char*)dy2 - (const char*)dy;
  if (plane < (long)n * h * w * lddy * 2 || plane + (double)n * h * w * lddy * 2.0 >= 2147483648.0) return DY_ERR_ARG;
  if (xs) {
    if (xs->nseg < 1 || xs->nseg > DY_MAX_SEGS || xs->c_end[xs->nseg - 1] != cin) return DY_ERR_ARG;
    for (int k = 0; k < xs->nseg; ++k)
      if ((xs->c_end[k] & 7) || (xs->ld[k] & 7) || !xs->ptr[k] || ((uintptr_t)xs->ptr[k] & 15) || xs->c_end[k] <= (k ? xs->c_end[k - 1] : 0) ||
          ((xs->acc[k] & 2) && ((h | w) & 1))) return DY_ERR_ARG;
  }
  WgBnHost bn{raw, draw, coef, acc, dgamma, dbeta, ldraw, count};
  bn.xs = xs;
  bn.dy_csplit = csplit; bn.dy_plane = (unsigned)plane;
  return conv_wgrad_impl(xs ? xs->ptr[0] : x, xs ? 8 : ldx, dy, lddy, slabs, dw, n, h, w, cin, cout, 1, 1, accumulate, 0, 0, 0, stream, &bn);
}
extern "C" int dy_conv_wgrad_ld_bn(const void* x, int ldx, const void* dy, int lddy, const void* raw, int ldraw, void* draw,
                                   const float* coef, const double* acc, float* dgamma, float* dbeta, float count, float* slabs,
                                   float* dw, int n, int h, int w, int cout, int ld_cin, int ld_taps, int ld_cphys, int accumulate,
                                   hipStream_t stream) {
  if (!raw || !coef || !acc || (ldraw & 7) || ((uintptr_t)raw & 15) || ((uintptr_t)draw & 15) || (cout & 15)) return DY_ERR_ARG;
  const WgBnHost bn{raw, draw, coef, acc, dgamma, dbeta, ldraw, count};
  return conv_wgrad_impl(x, ldx, dy, lddy, slabs, dw, n, h, w, ld_taps * ld_cphys, cout, 1, 1, accumulate, ld_taps, ld_cphys, ld_cin,
                         stream, &bn);
}

static int conv_wgrad_impl(const void* x, int ldx, const void* dy, int lddy, float* slabs, float* dw, int n, int h, int w,
                           int cin, int cout, int ks, int stride, int accumulate, int ld_taps, int ld_cphys, int ld_cin,
                           hipStream_t stream, const WgBnHost* bn) {
  if (!(ks == 1 || ks == 3) || !(stride == 1 || stride == 2) || (ks == 1 && stride != 1)) return DY_ERR_ARG;
  if ((ldx & 7) || (lddy & 7) || ((uintptr_t)x & 15) || ((uintptr_t)dy & 15)) return DY_ERR_ALIGN;
  // staging uses 32-bit buffer offsets: one image (3x3) or the whole tensor (1x1) must stay below 2 GiB
  if ((ks == 1 ? (double)n : 1.0) * h * w * (ldx > lddy ? ldx : lddy) * 2.0 >= 2147483648.0) return DY_ERR_ARG;
  int cp, op, nci, mtc;
  wgrad_geometry(cin, cout, ks, stride, &cp, &op, &nci, &mtc,
                 (long)n * ((h + 2 * (ks / 2) - ks) / stride + 1) * ((w + 2 * (ks / 2) - ks) / stride + 1));
  WgArgs a{};
  a.x = (const f16*)x; a.dy = (const f16*)dy; a.slabs = slabs; a.ldx = ldx; a.lddy = lddy;
  a.N = n; a.H = h; a.W = w;
  const int pad = ks / 2;
  a.Ho = (h + 2 * pad - ks) / stride + 1;
  a.Wo = (w + 2 * pad - ks) / stride + 1;
  a.cin_p = cp; a.cout_p = op;
  a.cin_r8 = (cin + 7) / 8 * 8; a.cout_r8 = (cout + 7) / 8 * 8;
  if (bn) {
    if ((ks == 1 ? (double)n : 1.0) * a.Ho * a.Wo * bn->ldraw * 2.0 >= 2147483648.0) return DY_ERR_ARG;
    a.raw = (const f16*)bn->raw; a.draw = (f16*)bn->draw; a.coef = bn->coef; a.acc = bn->acc;
    a.dgamma = bn->dgamma; a.dbeta = bn->dbeta; a.ldraw = bn->ldraw; a.count = bn->count;
    a.cout = bn->raw ? cout : (cout + 7) / 8 * 8;  // bias sums: one slot per physical channel of dY
    if (bn->xs) a.xs = *bn->xs;
    a.dy_csplit = bn->dy_csplit; a.dy_plane = bn->dy_plane;
  }
  a.nci_chunks = cp / (16 * nci);
  const int nco_chunks = op / (16 * mtc);
  const int th = wg_th(ks, stride, nci, mtc);
  if (ks == 1) {
    a.npix = (long)n * a.Ho * a.Wo;
    a.ntiles = (int)((a.npix + th * 32 - 1) / (th * 32));
  } else {
    a.tiles_x = cdiv(a.Wo, 32);
    a.tiles_y = cdiv(a.Ho, th);
    a.ntiles = a.tiles_x * a.tiles_y * n;
  }
  int nslabs;
  long slab_elems;
  dy_wgrad_workspace(n, h, w, cin, cout, ks, stride, &nslabs, &slab_elems);
  const int gy = a.nci_chunks * nco_chunks;
  int rc;
  if (ks == 1) rc = dispatch_wgrad<1, 1>(nci, mtc, a, nslabs, gy, stream);
  else if (stride == 1) rc = dispatch_wgrad<3, 1>(nci, mtc, a, nslabs, gy, stream);
  else rc = dispatch_wgrad<3, 2>(nci, mtc, a, nslabs, gy, stream);
  if (rc != DY_OK) return rc;
  if (!dw) return DY_OK;  // deferred: the caller reduces this layer's slabs later through dy_wgrad_reduce_batched
  const int total = ks * ks * op * cp;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 64)), dim3(256), 0, stream, slabs, nslabs, dw, cout, cin,
                     ks * ks, op, cp, accumulate, ld_taps, ld_cphys, ld_cin);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
