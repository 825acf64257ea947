// pa_core.hip -- context, level (BoxArray + owner map), MultiFab, ghost-cell kernels.
// gfx950 only.  Replaces the AMReX pieces the reference tools lean on for this path:
// MultiFab storage, FabArray::FillBoundary and MLCellLinOp::applyBC.
#include "pa_internal.h"
#include "pa_dist.h"
#include <algorithm>
#include <atomic>
#include <cstring>
#include <numeric>

int pa_fail(pa_ctx* ctx, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return 1;
}

extern "C" int pa_version(void) { return 100; }

// Workgroup table of a sweep (MarchArgs::wgtab, GradMarchArgs::wgtab), cached on the level per box class (0: wider than 32
// cells, 1: at most 32, 2: all) and tile shape.  Workgroup i runs on XCD i % 8, so the table is eight QUEUES: a box's tiles are
// cut into chunks of consecutive tiles (neighbouring tiles share halo rows / planes through that XCD's L2), the chunks are
// dealt largest first to the queue with the least work, and entry 8 s + q is the s-th tile of queue q ({-1, 0}: the queue has
// run out).  Round 5: chunks instead of whole boxes -- a level of a few large boxes (what pa_level_retile makes of a Pele
// BoxArray) or of a box count that is not a multiple of eight keeps all eight XCDs busy: 27 boxes of 128^3 ran at 0.54 of HBM
// against 0.61 for 64 of them, four boxes of 256 x 128 x 128 on four XCDs.  Null when the order-2 arithmetic (every box takes
// the tile count of the largest, boxes in groups of eight) launches at most 3 % more workgroups than the queues are long and
// the caller does not insist -- the regular tilings keep the launch they were tuned on.
static int host_owner(const pa_level* L, int i, int j, int k);
// part (round 6, sharded levels): 0 = every tile; 1 = the EARLY tiles, 2 = the rest.  A tile is early when every cell it reads
// outside its own box -- two cells around the tile, clipped to the FAB -- is a valid cell of a box THIS RANK owns: the local
// FillBoundary alone completes its input, so its sweep can start while the cross-rank exchange and the ghost preparation that
// follows it (coarse patches, special faces, ring) are still under way.  Tables of a part list tiles one by one (always built).
const WgTab* pa_sweep_wgtab(const pa_level* L, int cls, int tw, int mty, int kseg, bool force, int part) {
  if (part) force = true;
  const long long key = ((long long)cls << 56) | ((long long)part << 54) | ((long long)(force ? 1 : 0) << 52) | ((long long)tw << 40) | ((long long)mty << 24) | (long long)kseg;
  auto it = L->wgtabs.find(key);
  if (it != L->wgtabs.end()) return (it->second->d || part) ? it->second.get() : nullptr;
  std::unique_ptr<WgTab> T(new WgTab());
  struct Chunk { int n, box, t0; };
  std::vector<std::pair<int, int>> bt;  // (tiles, box)
  std::vector<std::vector<int>> sel;    // part != 0: the tiles of bt[i] that belong to the part
  long long real = 0;
  int tmax = 0;
  for (int b = 0; b < (int)L->boxes.size(); ++b) {
    const DBox& B = L->boxes[b];
    const int nx = B.hi[0] - B.lo[0] + 1, ny = B.hi[1] - B.lo[1] + 1, nz = B.hi[2] - B.lo[2] + 1;
    if (cls != 2 && (nx <= 32) != (cls == 1)) continue;
    const int tx = (nx + tw - 1) / tw, ty = (ny + mty - 1) / mty, tz = (nz + kseg - 1) / kseg;
    int t = tx * ty * tz;
    if (part) {
      std::vector<int> mine;
      for (int id = 0; id < t; ++id) {
        const int bx = id % tx, by = (id / tx) % ty, bz = id / (tx * ty);
        int lo[3] = {B.lo[0] + bx * tw, B.lo[1] + by * mty, B.lo[2] + bz * kseg}, hi[3];
        hi[0] = std::min(lo[0] + tw - 1, B.hi[0]); hi[1] = std::min(lo[1] + mty - 1, B.hi[1]); hi[2] = std::min(lo[2] + kseg - 1, B.hi[2]);
        bool early = true;
        for (int d = 0; d < 3 && early; ++d)
          for (int side = 0; side < 2 && early; ++side) {
            // the slab of the read region beyond face (d, side) of the box (empty when the tile does not come within two cells of it)
            int slo[3], shi[3];
            for (int q = 0; q < 3; ++q) { slo[q] = std::max(lo[q] - 2, B.lo[q] - 2); shi[q] = std::min(hi[q] + 2, B.hi[q] + 2); }
            if (side == 0) shi[d] = std::min(shi[d], B.lo[d] - 1); else slo[d] = std::max(slo[d], B.hi[d] + 1);
            if (slo[d] > shi[d]) continue;
            for (int k = slo[2]; k <= shi[2] && early; ++k)
              for (int j = slo[1]; j <= shi[1] && early; ++j)
                for (int i = slo[0]; i <= shi[0]; ++i)
                  if (host_owner(L, i, j, k) < 0) { early = false; break; }
          }
        if (early == (part == 1)) mine.push_back(id);
      }
      t = (int)mine.size();
      if (t == 0) continue;
      sel.push_back(std::move(mine));
    }
    bt.push_back({t, b});
    real += t;
    tmax = std::max(tmax, t);
  }
  const long long launched = (long long)tmax * 8 * (((long long)bt.size() + 7) / 8);
  if (!bt.empty()) {
    constexpr int chunk_env = 0;  // tiles per chunk (0: from the level's size)
    const int cap = chunk_env > 0 ? chunk_env : (int)std::max<long long>(8, std::min<long long>(64, real / 32));
    std::vector<Chunk> ch;
    for (size_t bi = 0; bi < bt.size(); ++bi) {
      const auto& p = bt[bi];
      const int parts = (p.first + cap - 1) / cap;
      for (int q = 0; q < parts; ++q) {
        const int a = (int)((long long)p.first * q / parts), e = (int)((long long)p.first * (q + 1) / parts);
        ch.push_back({e - a, part ? (int)bi : p.second, a});  // part: box = index into bt / sel, t0 = position in its tile list
      }
    }
    std::stable_sort(ch.begin(), ch.end(), [](const Chunk& a, const Chunk& b) { return a.n > b.n; });
    std::vector<Chunk> queue[8];
    long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const Chunk& c : ch) {
      int q = 0;
      for (int x = 1; x < 8; ++x)
        if (load[x] < load[q]) q = x;
      queue[q].push_back(c);
      load[q] += c.n;
    }
    const long long qlen = *std::max_element(load, load + 8);
    if (force || launched * 100 > qlen * 8 * 103) {
      std::vector<int> tab((size_t)(qlen * 8 * 2), 0);
      for (int q = 0; q < 8; ++q) {
        long long s = 0;
        for (const Chunk& c : queue[q])
          for (int t = 0; t < c.n; ++t, ++s) {
            tab[(size_t)(2 * (8 * s + q))] = part ? bt[(size_t)c.box].second : c.box;
            tab[(size_t)(2 * (8 * s + q) + 1)] = part ? sel[(size_t)c.box][(size_t)(c.t0 + t)] : c.t0 + t;
          }
        for (; s < qlen; ++s) tab[(size_t)(2 * (8 * s + q))] = -1;
      }
      if (hipMalloc(&T->d, sizeof(int) * tab.size()) == hipSuccess && hipMemcpy(T->d, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice) == hipSuccess) {
        T->n = (unsigned)(tab.size() / 2);
        T->h = std::move(tab);
      } else {
        if (T->d) (void)hipFree(T->d);
        T->d = nullptr;
        (void)hipGetLastError();  // no table: the order-2 launch does the same work
      }
    }
  }
  const WgTab* raw = T.get();
  L->wgtabs[key] = std::move(T);
  return (raw->d || part) ? raw : nullptr;  // a part's table may be empty (n == 0): the caller launches nothing for it
}

// (see pa_internal.h)  The three columns behind an x face of box B are read by: the boundary condition on G at that face and the
// fix-up of its first layer when the face is special; the FillBoundary of G into a neighbouring box B' across the face where B' has
// fix-up cells next to it -- B' has a special face on the same plane (the facing face, partly covered by B), or a special y / z face
// whose plane meets B's face away from B's own first three rows / planes (those are stored in any case).
const unsigned char* pa_sweep_gneed(const pa_level* L, const WgTab* T, int tw, int mty, int kseg, long long fine_serial, const std::vector<int>* cpregs) {
  if (!T || !T->d || T->h.size() != 2 * (size_t)T->n) return nullptr;
  auto it = T->gneed.find(fine_serial);
  if (it != T->gneed.end()) return it->second;
  const int nb = (int)L->boxes.size();
  pa_level* Lm = const_cast<pa_level*>(L);
  if ((int)Lm->xneed.size() != nb) {
    std::vector<unsigned char> sp((size_t)nb, 0);  // special faces of box b: bit 2 * dir + side
    for (int f : L->sfaces) sp[(size_t)(f / 6)] |= (unsigned char)(1u << (f % 6));
    Lm->xneed.assign((size_t)nb, 0);
    for (int b = 0; b < nb; ++b) {
      const DBox& B = L->boxes[b];
      for (int side = 0; side < 2; ++side) {
        bool need = (sp[(size_t)b] >> side) & 1;
        const int x = side ? B.hi[0] + 1 : B.lo[0] - 1;
        int last = -1;
        for (int k = B.lo[2]; k <= B.hi[2] && !need; k += L->g)
          for (int j = B.lo[1]; j <= B.hi[1] && !need; j += L->g) {
            const int o = host_owner(L, x, j, k);
            if (o < 0 || o == last) continue;
            last = o;
            const DBox& N = L->boxes[o];
            if ((sp[(size_t)o] >> (1 - side)) & 1) need = true;
            for (int d = 1; d < 3 && !need; ++d)
              for (int t = 0; t < 2; ++t) {
                const int c = t ? N.hi[d] : N.lo[d];
                if (((sp[(size_t)o] >> (2 * d + t)) & 1) && c >= B.lo[d] + 3 && c <= B.hi[d] - 3) need = true;
              }
          }
        if (need) Lm->xneed[(size_t)b] |= (unsigned char)(2u << side);
      }
    }
  }
  std::vector<std::vector<unsigned char>> full((size_t)nb);
  auto tiles = [&](const DBox& B, int& tx, int& ty, int& tz) {
    tx = (B.hi[0] - B.lo[0] + tw) / tw; ty = (B.hi[1] - B.lo[1] + mty) / mty; tz = (B.hi[2] - B.lo[2] + kseg) / kseg;
  };
  if (cpregs)
    for (size_t r = 0; r + 12 <= cpregs->size(); r += 12) {
      const int* R = cpregs->data() + r;
      const int sb = R[1], dir = R[9], t0 = dir == 0 ? 1 : 0, t1 = dir == 2 ? 1 : 2;
      if (sb < 0 || sb >= nb) continue;
      const DBox& B = L->boxes[sb];
      int lo[3] = {R[4], R[5], R[6]}, hi[3] = {R[4], R[5], R[6]};
      hi[t0] += R[7] - 1; hi[t1] += R[8] - 1;
      int tx, ty, tz;
      tiles(B, tx, ty, tz);
      auto& F = full[(size_t)sb];
      if (F.empty()) F.assign((size_t)tx * ty * tz, 0);
      const int a0 = std::max(0, (lo[0] - B.lo[0]) / tw), a1 = std::min(tx - 1, (hi[0] - B.lo[0]) / tw);
      const int b0 = std::max(0, (lo[1] - B.lo[1]) / mty), b1 = std::min(ty - 1, (hi[1] - B.lo[1]) / mty);
      const int c0 = std::max(0, (lo[2] - B.lo[2]) / kseg), c1 = std::min(tz - 1, (hi[2] - B.lo[2]) / kseg);
      for (int c = c0; c <= c1; ++c)
        for (int bb = b0; bb <= b1; ++bb)
          for (int a = a0; a <= a1; ++a) F[((size_t)c * ty + bb) * tx + a] = 1;
    }
  std::vector<unsigned char> need((size_t)T->n, 0);
  for (unsigned e = 0; e < T->n; ++e) {
    const int b = T->h[2 * (size_t)e], id = T->h[2 * (size_t)e + 1];
    if (b < 0 || b >= nb) continue;
    unsigned char v = Lm->xneed[(size_t)b];
    if (!full[(size_t)b].empty() && id >= 0 && (size_t)id < full[(size_t)b].size() && full[(size_t)b][(size_t)id]) v |= 1;
    need[(size_t)e] = v;
  }
  unsigned char* d = nullptr;
  if (hipMalloc(&d, need.size()) != hipSuccess || hipMemcpy(d, need.data(), need.size(), hipMemcpyHostToDevice) != hipSuccess) {
    if (d) (void)hipFree(d);
    (void)hipGetLastError();
    return nullptr;
  }
  T->gneed[fine_serial] = d;
  return d;
}

static pa_options g_opt;
static bool g_opt_read = false;
extern "C" void pa_options_reload(void) {
  pa_options o;
  auto geti = [](const char* n, int def) { const char* e = getenv(n); return e ? atoi(e) : def; };
  o.filter_exact = geti("PA_FILTER_EXACT", 0) != 0;
  o.allow_unverified_gaussian = geti("PA_ALLOW_UNVERIFIED_GAUSSIAN", 0) != 0;
  if (const char* e = getenv("PA_RETILE_MAX")) {
    int v[3] = {0, 0, 0};
    if (sscanf(e, "%d %d %d", &v[0], &v[1], &v[2]) == 3 && v[0] > 0 && v[1] > 0 && v[2] > 0)
      for (int d = 0; d < 3; ++d) o.retile_max[d] = v[d];
  }
  o.fused2 = geti("PA_FUSED2", 1) != 0;
  o.ncg = geti("PA_NCG", 1) != 0;
  o.dist_early = geti("PA_DIST_EARLY", 0) != 0;
  o.smooth_replicated = geti("PA_SMOOTH_REPLICATED", 0) != 0;
  o.smooth_mg = geti("PA_SMOOTH_MG", -1);
  o.smooth_march = geti("PA_SMOOTH_MARCH", 1) != 0;
  o.smooth_timing = geti("PA_SMOOTH_TIMING", 0) != 0;
  o.force_fallbacks = geti("PA_FORCE_FALLBACKS", 0) != 0;
  o.scratch_poison = geti("PA_SCRATCH_POISON", 0) != 0;
  g_opt = o;
  g_opt_read = true;
}
const pa_options& pa_opt() {
  if (!g_opt_read) pa_options_reload();
  return g_opt;
}

extern "C" pa_ctx* pa_ctx_create(int device, void* hip_stream) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  pa_ctx* ctx = new pa_ctx();
  ctx->device = device;
  if (hip_stream) {
    ctx->stream = (hipStream_t)hip_stream;
  } else {
    if (hipStreamCreate(&ctx->stream) != hipSuccess) { delete ctx; return nullptr; }
    ctx->own_stream = true;
  }
  if (hipMalloc(&ctx->d_flags, 16 * sizeof(int)) != hipSuccess || hipMemset(ctx->d_flags, 0, 16 * sizeof(int)) != hipSuccess) {
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return nullptr;
  }
  // the side stream of the pipelines (boundary kernels next to a sweep, exchanges next to the local ghost fill): creating a
  // stream costs ~10 ms the first time, so it is made here -- the tools bring the context up while they read the plotfile --
  // and not inside the first pass.  A failure leaves it to the first use.
  if (hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess) { ctx->stream2 = nullptr; (void)hipGetLastError(); }
  return ctx;
}

extern "C" void pa_ctx_destroy(pa_ctx* ctx) {
  PaBind bind_(ctx);
  if (!ctx) return;
  pa_rccl_destroy(ctx);
  if (ctx->d_red) (void)hipFree(ctx->d_red);
  if (ctx->d_flags) (void)hipFree(ctx->d_flags);
  if (ctx->d_slow) (void)hipFree(ctx->d_slow);
  if (ctx->d_prog) (void)hipFree(ctx->d_prog);
  for (hipEvent_t e : ctx->fix_evs)
    if (e) (void)hipEventDestroy(e);
  if (ctx->d_scr) (void)hipFree(ctx->d_scr);
  if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
  if (ctx->d_mcz) (void)hipFree(ctx->d_mcz);
  for (auto& c : ctx->surf_cache) (void)hipFree(c.first);
  for (auto& e : ctx->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto& e : ctx->sync_evs) (void)hipEventDestroy(e);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  for (auto& st : ctx->lev_streams) (void)hipStreamDestroy(st);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int pa_profile_enable(pa_ctx* ctx, int on) {
  PaBind bind_(ctx);
  if (!ctx) return 1;
  // 0: off, 1: every tag, otherwise a bit mask (1 << tag): e.g. 2 = the fused sweep only, so that a timed region pays for
  // two events per sweep launch and nothing else (the events of ~20 small boundary launches per step cost ~2 % of it)
  ctx->profile = on == 0 ? 0u : (on == 1 ? 0xffffffffu : (unsigned)on);
  return 0;
}

// sum of the durations (ms) of all launches recorded under `tag` since the last reset; synchronous
extern "C" int pa_profile_read(pa_ctx* ctx, int tag, int64_t* nlaunch, double* total_ms, int reset) {
  PaBind bind_(ctx);
  if (!ctx || !nlaunch || !total_ms) return 1;
  PA_HIP(hipStreamSynchronize(ctx->stream));
  *nlaunch = 0;
  *total_ms = 0.0;
  for (auto& e : ctx->evs) {
    if (e.tag != tag) continue;
    float ms = 0.f;
    PA_HIP(hipEventElapsedTime(&ms, e.a, e.b));
    *total_ms += ms;
    ++*nlaunch;
  }
  if (reset) {
    for (auto& e : ctx->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    ctx->evs.clear();
  }
  return 0;
}

extern "C" const char* pa_sweep_kernel_name(const pa_ctx* ctx) { return ctx ? ctx->sweep_kernel.c_str() : ""; }
extern "C" const char* pa_last_error(const pa_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
extern "C" void* pa_ctx_stream(pa_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int pa_sync(pa_ctx* ctx) {
  PaBind bind_(ctx);
  if (!ctx) return 1;
  PA_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// ------------------------------------------------------------------------ level
static int host_classify(const pa_level* L, int i, int j, int k) {
  int p[3] = {i, j, k};
  for (int d = 0; d < 3; ++d) {
    const int len = L->domhi[d] - L->domlo[d] + 1;
    if (p[d] < L->domlo[d] || p[d] > L->domhi[d]) {
      if (!L->is_per[d]) return 2;
      while (p[d] < L->domlo[d]) p[d] += len;
      while (p[d] > L->domhi[d]) p[d] -= len;
    }
  }
  int m[3];
  for (int d = 0; d < 3; ++d) {
    const int r = p[d] - L->mlo[d];
    if (r < 0) return 1;
    m[d] = r / L->g;
    if (m[d] >= L->mn[d]) return 1;
  }
  const int o = L->owner[((size_t)m[2] * L->mn[1] + m[1]) * L->mn[0] + m[0]];
  return (o != -1) ? 0 : 1;  // <= -2: valid cell of a box owned by another rank
}

// the owner-map entry of a cell (through periodic images): >= 0 a box of this rank, <= -2 a box of another rank, -1 none / outside a wall
static int host_owner(const pa_level* L, int i, int j, int k) {
  int p[3] = {i, j, k};
  for (int d = 0; d < 3; ++d) {
    const int len = L->domhi[d] - L->domlo[d] + 1;
    if (p[d] < L->domlo[d] || p[d] > L->domhi[d]) {
      if (!L->is_per[d]) return -1;
      while (p[d] < L->domlo[d]) p[d] += len;
      while (p[d] > L->domhi[d]) p[d] -= len;
    }
  }
  int m[3];
  for (int d = 0; d < 3; ++d) {
    const int r = p[d] - L->mlo[d];
    if (r < 0) return -1;
    m[d] = r / L->g;
    if (m[d] >= L->mn[d]) return -1;
  }
  return L->owner[((size_t)m[2] * L->mn[1] + m[1]) * L->mn[0] + m[0]];
}

// true if face (d, side) of box B (any box of the level, local or not) has a ghost cell that is not a valid cell of the
// level.  The class of a ghost cell is constant over a g-block of the owner map in the tangential directions, so one
// probe per block is enough.
bool pa_face_is_special(const pa_level* L, const DBox& B, int d, int side) {
  const int t0 = (d == 0) ? 1 : 0, t1 = (d == 2) ? 1 : 2;
  int q[3];
  q[d] = side ? B.hi[d] + 1 : B.lo[d] - 1;
  for (int v = B.lo[t1]; v <= B.hi[t1]; v += L->g)
    for (int u = B.lo[t0]; u <= B.hi[t0]; u += L->g) {
      q[t0] = u; q[t1] = v;
      if (host_classify(L, q[0], q[1], q[2]) != 0) return true;
    }
  return false;
}
int pa_host_classify(const pa_level* L, int i, int j, int k) { return host_classify(L, i, j, k); }

// cf_masks of every ghost cell of every special face (once per level; ratio 2)
__global__ __launch_bounds__(256) void k_build_sfcode(DLevelView L, unsigned short* code) {
  int b, dir, side, layer, q[3];
  DBox B;
  const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (!sface_decode(L, blockIdx.y, t, 1, b, B, dir, side, q, layer)) return;
  code[L.sfoff[blockIdx.y] + t] = (unsigned short)cf_masks(L, q, dir, 2);
}

// flags of the chunk records (SfChunk): one workgroup per chunk, thread = 2 x 2 block as in the kernels that use them
__global__ __launch_bounds__(256) void k_sfchunk_flags(DLevelView L, SfChunk* ck) {
  SfChunk& D = ck[blockIdx.x];
  const int dir = D.dir_side >> 1, t0 = dir == 0 ? 1 : 0, t1 = dir == 2 ? 1 : 2;
  const int n0 = D.hi[t0] - D.lo[t0] + 1, n1 = D.hi[t1] - D.lo[t1] + 1;
  const int hw = D.cw >> 1, bu = (int)threadIdx.x % hw, bv = (int)threadIdx.x / hw;
  bool full = true, wall = true, cf = false, valid = false;
  for (int dv = 0; dv < 2; ++dv)
    for (int du = 0; du < 2; ++du) {
      const int u = D.u0 + 2 * bu + du, v = D.v0 + 2 * bv + dv;
      if (u < 0 || v < 0 || u >= n0 || v >= n1) continue;
      const unsigned code = L.sfcode[D.sfoff + (long long)v * n0 + u];
      full = full && code == PA_CODE_FULL;
      wall = wall && (code & 3u) == 2u;
      cf = cf || (code & 3u) == 1u;
      valid = valid || (code & 3u) == 0u;
    }
  const int afull = __syncthreads_and(full ? 1 : 0), awall = __syncthreads_and(wall ? 1 : 0);
  const int acf = __syncthreads_or(cf ? 1 : 0), avalid = __syncthreads_or(valid ? 1 : 0);
  if (threadIdx.x == 0) {
    // the straight-line kinds want whole blocks: even start, even extents
    const bool even = !((D.lo[t0] | D.lo[t1] | n0 | n1 | D.u0 | D.v0) & 1);
    D.flags = (afull && even && D.cpoff >= 0 ? PA_SFC_FULL : 0) | (awall && even ? PA_SFC_WALL : 0) | (acf ? PA_SFC_HAS_CF : 0) | (avalid ? PA_SFC_HAS_VALID : 0);
  }
}

// Everything a level is made from.  Unsharded: gboxes empty, every box local.  Sharded: `local` are the global boxes
// owned by `rank` (global order, gid = their global indices), the other global boxes only mark their cells as valid
// cells of the level (owner-map entry -2 - g).
pa_level* pa_level_create_spec(pa_ctx* ctx, const LevelSpec& S, const int32_t domlo[3], const int32_t domhi[3], const int32_t is_per[3],
                               const double prob_lo[3], const double prob_hi[3]);

static bool read_boxes(pa_ctx* ctx, int n, const int32_t* b6, const int32_t domlo[3], const int32_t domhi[3], std::vector<DBox>& out, const char* who) {
  out.resize(n);
  for (int b = 0; b < n; ++b)
    for (int d = 0; d < 3; ++d) {
      out[b].lo[d] = b6[6 * b + d];
      out[b].hi[d] = b6[6 * b + 3 + d];
      if (out[b].hi[d] < out[b].lo[d] || out[b].lo[d] < domlo[d] || out[b].hi[d] > domhi[d]) {
        pa_fail(ctx, std::string(who) + ": box " + std::to_string(b) + " is empty or outside the domain");
        return false;
      }
    }
  return true;
}

extern "C" pa_level* pa_level_create(pa_ctx* ctx, int nboxes, const int32_t* b6, const int32_t domlo[3],
                                     const int32_t domhi[3], const int32_t is_per[3], const double prob_lo[3],
                                     const double prob_hi[3]) {
  PaBind bind_(ctx);
  if (!ctx) return nullptr;
  if (nboxes <= 0 || !b6) { pa_fail(ctx, "pa_level_create: empty BoxArray"); return nullptr; }
  LevelSpec S;
  if (!read_boxes(ctx, nboxes, b6, domlo, domhi, S.local, "pa_level_create")) return nullptr;
  return pa_level_create_spec(ctx, S, domlo, domhi, is_per, prob_lo, prob_hi);
}

// One rank's share of a level (replaces DistributionMapping(ba), grad.cpp:162 / curvature.cpp:289): the whole BoxArray,
// the owner rank of every box, and which rank this context is.  A rank may own no box of a level.
extern "C" pa_level* pa_level_create_sharded(pa_ctx* ctx, int nboxes, const int32_t* b6, const int32_t* owner, int rank, int nranks,
                                             const int32_t domlo[3], const int32_t domhi[3], const int32_t is_per[3], const double prob_lo[3],
                                             const double prob_hi[3]) {
  PaBind bind_(ctx);
  if (!ctx) return nullptr;
  if (nboxes <= 0 || !b6 || !owner) { pa_fail(ctx, "pa_level_create_sharded: empty BoxArray"); return nullptr; }
  if (nranks < 1 || rank < 0 || rank >= nranks) { pa_fail(ctx, "pa_level_create_sharded: bad rank / nranks"); return nullptr; }
  LevelSpec S;
  S.rank = rank; S.nranks = nranks;
  if (!read_boxes(ctx, nboxes, b6, domlo, domhi, S.gboxes, "pa_level_create_sharded")) return nullptr;
  S.gowner.assign(owner, owner + nboxes);
  for (int g = 0; g < nboxes; ++g) {
    if (owner[g] < 0 || owner[g] >= nranks) { pa_fail(ctx, "pa_level_create_sharded: owner of box " + std::to_string(g) + " out of range"); return nullptr; }
    if (owner[g] == rank) { S.local.push_back(S.gboxes[g]); S.gid.push_back(g); }
  }
  return pa_level_create_spec(ctx, S, domlo, domhi, is_per, prob_lo, prob_hi);
}

static std::atomic<long long> g_level_serial{0};

pa_level* pa_level_create_spec(pa_ctx* ctx, const LevelSpec& S, const int32_t domlo[3], const int32_t domhi[3], const int32_t is_per[3],
                               const double prob_lo[3], const double prob_hi[3]) {
  if (!ctx) return nullptr;
  const int nboxes = (int)S.local.size();
  pa_level* L = new pa_level();
  L->serial = ++g_level_serial;
  L->ctx = ctx;
  L->boxes = S.local;
  L->rank = S.rank; L->nranks = S.nranks;
  L->gboxes = S.gboxes; L->gowner = S.gowner; L->gid = S.gid;
  L->source_only = S.source_only;
  L->glocal.assign(S.gboxes.size(), -1);
  for (size_t b = 0; b < S.gid.size(); ++b) L->glocal[S.gid[b]] = (int)b;
  // boxes of the level owned by other ranks
  std::vector<DBox> remote;
  std::vector<int> remote_gid;
  for (size_t g = 0; g < S.gboxes.size(); ++g)
    if (S.gowner[g] != S.rank) { remote.push_back(S.gboxes[g]); remote_gid.push_back((int)g); }
  const int nremote = (int)remote.size();
  L->nremote = nremote;
  for (int d = 0; d < 3; ++d) {
    L->domlo[d] = domlo[d]; L->domhi[d] = domhi[d]; L->is_per[d] = is_per[d] ? 1 : 0;
    L->prob_lo[d] = prob_lo[d]; L->prob_hi[d] = prob_hi[d];
    // Geometry: dx = length / ncells ; inv_dx = 1/dx
    L->dx[d] = (prob_hi[d] - prob_lo[d]) / (double)(domhi[d] - domlo[d] + 1);
    L->dxinv[d] = 1.0 / L->dx[d];
    L->mlo[d] = INT32_MAX;
  }
  int mhi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  for (int b = 0; b < nboxes + nremote; ++b) {
    const DBox& B = b < nboxes ? L->boxes[b] : remote[b - nboxes];
    for (int d = 0; d < 3; ++d) {
      L->mlo[d] = std::min(L->mlo[d], B.lo[d]);
      mhi[d] = std::max(mhi[d], B.hi[d]);
      if (b < nboxes) L->maxn[d] = std::max(L->maxn[d], B.hi[d] - B.lo[d] + 1);
    }
    if (b < nboxes) L->ncells += (long long)(B.hi[0] - B.lo[0] + 1) * (B.hi[1] - B.lo[1] + 1) * (B.hi[2] - B.lo[2] + 1);
  }
  // owner-map granularity
  int g = 0;
  for (int b = 0; b < nboxes + nremote; ++b) {
    const DBox& B = b < nboxes ? L->boxes[b] : remote[b - nboxes];
    for (int d = 0; d < 3; ++d) {
      g = std::gcd(g, B.lo[d] - L->mlo[d]);
      g = std::gcd(g, B.hi[d] - B.lo[d] + 1);
    }
  }
  if (g <= 0) g = 1;
  L->g = g;
  size_t msz = 1;
  for (int d = 0; d < 3; ++d) {
    L->mn[d] = (mhi[d] - L->mlo[d] + 1) / g;
    msz *= (size_t)L->mn[d];
  }
  if (msz > ((size_t)1 << 28)) {
    pa_fail(ctx, "pa_level_create: owner map too large (boxes not aligned to a common blocking factor)");
    delete L;
    return nullptr;
  }
  L->owner.assign(msz, -1);
  for (int b = 0; b < nboxes + nremote; ++b) {
    const DBox& B = b < nboxes ? L->boxes[b] : remote[b - nboxes];
    for (int kz = (B.lo[2] - L->mlo[2]) / g; kz <= (B.hi[2] - L->mlo[2]) / g; ++kz)
      for (int ky = (B.lo[1] - L->mlo[1]) / g; ky <= (B.hi[1] - L->mlo[1]) / g; ++ky)
        for (int kx = (B.lo[0] - L->mlo[0]) / g; kx <= (B.hi[0] - L->mlo[0]) / g; ++kx) {
          int& o = L->owner[((size_t)kz * L->mn[1] + ky) * L->mn[0] + kx];
          if (o != -1) {
            pa_fail(ctx, "pa_level_create: boxes " + std::to_string(o) + " and " + std::to_string(b) + " overlap");
            delete L;
            return nullptr;
          }
          o = b < nboxes ? b : -2 - remote_gid[b - nboxes];
        }
  }
  // Fused grad->curvature legality: an edge ghost cell that is NOT a valid cell while both
  // of its face-ring neighbours towards the box ARE valid cells (concave coarse-fine corner)
  // would need two different boundary values in one FAB slot.  A property of the whole BoxArray:
  // every rank of a sharded level must take the same path (the paths exchange different data).
  L->fusable = true;
  const int nfus = S.source_only ? 0 : (S.gboxes.empty() ? nboxes : (int)S.gboxes.size());
  for (int b = 0; b < nfus && L->fusable; ++b) {
    const DBox& B = S.gboxes.empty() ? L->boxes[b] : S.gboxes[b];
    for (int a = 0; a < 3 && L->fusable; ++a)
      for (int c = a + 1; c < 3 && L->fusable; ++c) {
        const int e = 3 - a - c;
        for (int sa = 0; sa < 2 && L->fusable; ++sa)
          for (int sc = 0; sc < 2 && L->fusable; ++sc)
            for (int t = B.lo[e]; t <= B.hi[e]; ++t) {
              int q[3];
              q[e] = t;
              q[a] = sa ? B.hi[a] + 1 : B.lo[a] - 1;
              q[c] = sc ? B.hi[c] + 1 : B.lo[c] - 1;
              if (host_classify(L, q[0], q[1], q[2]) == 0) continue;
              int qa[3] = {q[0], q[1], q[2]}, qc[3] = {q[0], q[1], q[2]};
              qa[a] += sa ? -1 : 1;
              qc[c] += sc ? -1 : 1;
              if (host_classify(L, qa[0], qa[1], qa[2]) == 0 && host_classify(L, qc[0], qc[1], qc[2]) == 0) {
                L->fusable = false;
                break;
              }
            }
      }
  }
  // Special faces: box faces with at least one adjacent ghost cell that is not a valid cell of the
  // level (coarse-fine or wall).
  for (int b = 0; b < nboxes && !S.source_only; ++b)
    for (int d = 0; d < 3; ++d)
      for (int side = 0; side < 2; ++side)
        if (pa_face_is_special(L, L->boxes[b], d, side)) L->sfaces.push_back(b * 6 + d * 2 + side);
  // Pure special faces (the exact-normal sweep of pa_fused2.hip wants them): a face with a ghost cell that is not a valid
  // cell has NO ghost cell that is one.  Again a property of the whole BoxArray (all ranks take the same path).
  L->pure_faces = true;
  for (int b = 0; b < nfus && L->pure_faces; ++b) {
    const DBox& B = S.gboxes.empty() ? L->boxes[b] : S.gboxes[b];
    for (int d = 0; d < 3 && L->pure_faces; ++d)
      for (int side = 0; side < 2 && L->pure_faces; ++side) {
        const int t0 = (d == 0) ? 1 : 0, t1 = (d == 2) ? 1 : 2;
        int q[3], nval = 0, nnot = 0;
        q[d] = side ? B.hi[d] + 1 : B.lo[d] - 1;
        for (int v = B.lo[t1]; v <= B.hi[t1]; v += g)
          for (int u = B.lo[t0]; u <= B.hi[t0]; u += g) {
            q[t0] = u; q[t1] = v;
            (host_classify(L, q[0], q[1], q[2]) == 0 ? nval : nnot)++;
          }
        if (nval && nnot) L->pure_faces = false;
      }
  }
  const size_t nsf_alloc = std::max<size_t>(L->sfaces.size(), 1);
  std::vector<int> sfindex((size_t)nboxes * 6, -1);
  std::vector<long long> sfoff(nsf_alloc, 0);
  long long ncode = 0;
  L->cgoff.assign(nsf_alloc, 0);
  L->cg_total = 0;
  std::vector<long long> cpoff(nsf_alloc, -1);  // coarse patches (DLevelView::cp): faces inside the domain or behind a periodic side
  L->cp_total = 0;
  for (size_t e = 0; e < L->sfaces.size(); ++e) {
    const int f = L->sfaces[e];
    const DBox& B = L->boxes[f / 6];
    const int d = (f % 6) >> 1, t0 = (d == 0) ? 1 : 0, t1 = (d == 2) ? 1 : 2;
    {
      const int side = f & 1, qd = side ? B.hi[d] + 1 : B.lo[d] - 1;
      if (L->is_per[d] || (qd >= L->domlo[d] && qd <= L->domhi[d])) {
        int plane, u0, v0, pw, ph;
        cpatch_geom(B, d, side, plane, u0, v0, pw, ph);
        cpoff[e] = L->cp_total;
        L->cp_total += ((long long)pw * ph + 7) / 8 * 8;
      }
    }
    sfindex[f] = (int)e;
    sfoff[e] = ncode;
    ncode += (long long)(B.hi[t0] - B.lo[t0] + 1) * (B.hi[t1] - B.lo[t1] + 1);
    L->cgoff[e] = L->cg_total;  // (n0+2) x (n1+2) + two rows of slack, starts on a 64-byte boundary
    L->cg_total += ((long long)(B.hi[t0] - B.lo[t0] + 3) * (B.hi[t1] - B.lo[t1] + 5) + 7) / 8 * 8;
  }
  {  // sweep groups by box width
    std::vector<int> wide, narrow;
    for (int b = 0; b < nboxes; ++b) {
      const DBox& B = L->boxes[b];
      const bool nar = B.hi[0] - B.lo[0] + 1 <= 32;
      (nar ? narrow : wide).push_back(b);
      for (int d = 0; d < 3; ++d) {
        int& m = nar ? L->nmax[d] : L->wmax[d];
        m = std::max(m, B.hi[d] - B.lo[d] + 1);
      }
    }
    L->nwide = (int)wide.size();
    L->nnarrow = (int)narrow.size();
    if (L->nwide && L->nnarrow) {
      wide.insert(wide.end(), narrow.begin(), narrow.end());
      if (hipMalloc(&L->d_blist, sizeof(int) * wide.size()) != hipSuccess ||
          hipMemcpy(L->d_blist, wide.data(), sizeof(int) * wide.size(), hipMemcpyHostToDevice) != hipSuccess) {
        pa_fail(ctx, "pa_level_create: device allocation failed");
        delete L;
        return nullptr;
      }
    }
  }
  {  // work table of the kernels that run one thread per ghost cell of a special face: faces differ in size by 16x on general
     // BoxArrays (32^2 .. 128^2 cells), a grid of (largest face / 256) x faces would be mostly empty workgroups
    std::vector<int> wg, pwg;
    for (size_t e = 0; e < L->sfaces.size(); ++e) {
      const int f = L->sfaces[e];
      const DBox& B = L->boxes[f / 6];
      const int d = (f % 6) >> 1, t0 = (d == 0) ? 1 : 0, t1 = (d == 2) ? 1 : 2;
      const long long nc = (long long)(B.hi[t0] - B.lo[t0] + 1) * (B.hi[t1] - B.lo[t1] + 1);
      for (int c = 0; c < (int)((nc + 255) / 256); ++c) { wg.push_back((int)e); wg.push_back(c); }
      // perimeter cells of the face as faces_curv_cell enumerates them: two full rows, then the end columns of the rows between
      const long long n0 = B.hi[t0] - B.lo[t0] + 1, n1 = B.hi[t1] - B.lo[t1] + 1, P = n1 >= 2 ? 2 * n0 + 2 * (n1 - 2) : n0;
      for (int c = 0; c < (int)((P + 255) / 256); ++c) { pwg.push_back((int)e); pwg.push_back(c); }
    }
    L->npfwg = (int)(pwg.size() / 2);
    if (L->npfwg > 0 && (hipMalloc(&L->d_pfwg, sizeof(int) * pwg.size()) != hipSuccess ||
                         hipMemcpy(L->d_pfwg, pwg.data(), sizeof(int) * pwg.size(), hipMemcpyHostToDevice) != hipSuccess)) {
      pa_fail(ctx, "pa_level_create: device allocation failed");
      delete L;
      return nullptr;
    }
    std::vector<int> sfb;
    for (int f : L->sfaces)
      if (sfb.empty() || sfb.back() != f / 6) sfb.push_back(f / 6);  // sfaces is sorted by box
    L->nsfboxes = (int)sfb.size();
    if (L->nsfboxes > 0 && (hipMalloc(&L->d_sfboxes, sizeof(int) * sfb.size()) != hipSuccess ||
                            hipMemcpy(L->d_sfboxes, sfb.data(), sizeof(int) * sfb.size(), hipMemcpyHostToDevice) != hipSuccess)) {
      pa_fail(ctx, "pa_level_create: device allocation failed");
      delete L;
      return nullptr;
    }
    {  // chunk records (SfChunk): rectangles of cw x ch = 1024 ghost cells, cw = the power of two that covers the face's rows (32 .. 512)
      std::vector<SfChunk> ck;
      for (size_t e = 0; e < L->sfaces.size(); ++e) {
        const int f = L->sfaces[e];
        const DBox& B = L->boxes[f / 6];
        const int d = (f % 6) >> 1, t0 = (d == 0) ? 1 : 0, t1 = (d == 2) ? 1 : 2;
        const int n0 = B.hi[t0] - B.lo[t0] + 1, n1 = B.hi[t1] - B.lo[t1] + 1;
        int cw = 32;
        while (cw < 512 && cw < n0) cw *= 2;
        const int ch = 1024 / cw;
        for (int v0 = -(B.lo[t1] & 1); v0 < n1; v0 += ch)  // blocks on even global indices
          for (int u0 = -(B.lo[t0] & 1); u0 < n0; u0 += cw) {
            SfChunk D;
            D.face = (int)e; D.box = f / 6; D.dir_side = f % 6; D.flags = 0;
            for (int q = 0; q < 3; ++q) { D.lo[q] = B.lo[q]; D.hi[q] = B.hi[q]; }
            D.u0 = u0; D.v0 = v0; D.cw = cw; D.ch = ch;
            D.sfoff = sfoff[e]; D.cgoff = L->cgoff[e]; D.cpoff = cpoff[e];
            ck.push_back(D);
          }
      }
      L->nsfchunk = (int)ck.size();
      if (L->nsfchunk > 0 && (hipMalloc(&L->d_sfchunk, sizeof(SfChunk) * ck.size()) != hipSuccess ||
                              hipMemcpy(L->d_sfchunk, ck.data(), sizeof(SfChunk) * ck.size(), hipMemcpyHostToDevice) != hipSuccess)) {
        pa_fail(ctx, "pa_level_create: device allocation failed");
        delete L;
        return nullptr;
      }
    }
    L->nsfwg = (int)(wg.size() / 2);
    if (L->nsfwg > 0 && (hipMalloc(&L->d_sfwg, sizeof(int) * wg.size()) != hipSuccess ||
                         hipMemcpy(L->d_sfwg, wg.data(), sizeof(int) * wg.size(), hipMemcpyHostToDevice) != hipSuccess)) {
      pa_fail(ctx, "pa_level_create: device allocation failed");
      delete L;
      return nullptr;
    }
  }
  if (hipMalloc(&L->d_boxes, sizeof(DBox) * std::max(nboxes, 1)) != hipSuccess ||
      hipMalloc(&L->d_sfaces, sizeof(int) * nsf_alloc) != hipSuccess ||
      (!L->sfaces.empty() && hipMemcpy(L->d_sfaces, L->sfaces.data(), sizeof(int) * L->sfaces.size(), hipMemcpyHostToDevice) != hipSuccess) ||
      hipMalloc(&L->d_sfindex, sizeof(int) * std::max<size_t>(sfindex.size(), 1)) != hipSuccess ||
      (!sfindex.empty() && hipMemcpy(L->d_sfindex, sfindex.data(), sizeof(int) * sfindex.size(), hipMemcpyHostToDevice) != hipSuccess) ||
      hipMalloc(&L->d_cpoff, sizeof(long long) * nsf_alloc) != hipSuccess ||
      hipMemcpy(L->d_cpoff, cpoff.data(), sizeof(long long) * nsf_alloc, hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc(&L->d_cgoff, sizeof(long long) * nsf_alloc) != hipSuccess ||
      hipMemcpy(L->d_cgoff, L->cgoff.data(), sizeof(long long) * nsf_alloc, hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc(&L->d_sfoff, sizeof(long long) * nsf_alloc) != hipSuccess ||
      hipMemcpy(L->d_sfoff, sfoff.data(), sizeof(long long) * nsf_alloc, hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc(&L->d_sfcode, sizeof(unsigned short) * std::max<long long>(ncode, 1)) != hipSuccess ||
      hipMalloc(&L->d_owner, sizeof(int) * msz) != hipSuccess ||
      (nboxes > 0 && hipMemcpy(L->d_boxes, L->boxes.data(), sizeof(DBox) * nboxes, hipMemcpyHostToDevice) != hipSuccess) ||
      hipMemcpy(L->d_owner, L->owner.data(), sizeof(int) * msz, hipMemcpyHostToDevice) != hipSuccess) {
    pa_fail(ctx, "pa_level_create: device allocation failed");
    if (L->d_boxes) (void)hipFree(L->d_boxes);
    if (L->d_owner) (void)hipFree(L->d_owner);
    if (L->d_sfaces) (void)hipFree(L->d_sfaces);
    if (L->d_sfindex) (void)hipFree(L->d_sfindex);
    if (L->d_sfoff) (void)hipFree(L->d_sfoff);
    if (L->d_cgoff) (void)hipFree(L->d_cgoff);
    if (L->d_cpoff) (void)hipFree(L->d_cpoff);
    if (L->d_sfcode) (void)hipFree(L->d_sfcode);
    delete L;
    return nullptr;
  }
  DLevelView& V = L->view;
  V.cgoff = L->d_cgoff;
  V.cg = nullptr;
  V.cpoff = L->d_cpoff;
  V.cp = nullptr;
  V.sfindex = L->d_sfindex;
  V.sfoff = L->d_sfoff;
  V.sfcode = L->d_sfcode;
  V.gshift = (g & (g - 1)) == 0 ? __builtin_ctz((unsigned)g) : -1;
  V.nsf = (int)L->sfaces.size();
  V.sfaces = L->d_sfaces;
  V.nboxes = nboxes;
  V.boxes = L->d_boxes;
  V.owner = L->d_owner;
  V.g = g;
  for (int d = 0; d < 3; ++d) {
    V.domlo[d] = L->domlo[d]; V.domhi[d] = L->domhi[d]; V.is_per[d] = L->is_per[d];
    V.mlo[d] = L->mlo[d]; V.mn[d] = L->mn[d]; V.dxinv[d] = L->dxinv[d];
  }
  if (!L->sfaces.empty()) {
    const long long n0 = L->maxn[0], n1 = L->maxn[1], n2 = L->maxn[2];
    const long long nt = std::max(n1 * n2, std::max(n0 * n2, n0 * n1));
    hipLaunchKernelGGL(k_build_sfcode, dim3((unsigned)((nt + 255) / 256), (unsigned)L->sfaces.size()), dim3(256), 0, ctx->stream, V, L->d_sfcode);
    if (L->nsfchunk > 0) hipLaunchKernelGGL(k_sfchunk_flags, dim3((unsigned)L->nsfchunk), dim3(256), 0, ctx->stream, V, L->d_sfchunk);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
      pa_fail(ctx, "pa_level_create: building the boundary masks failed");
      pa_level_destroy(L);
      return nullptr;
    }
  }
  return L;
}

extern "C" void pa_level_destroy(pa_level* L) {
  if (!L) return;
  if (L->d_sfaces) (void)hipFree(L->d_sfaces);
  if (L->d_sfindex) (void)hipFree(L->d_sfindex);
  if (L->d_sfoff) (void)hipFree(L->d_sfoff);
  if (L->d_cgoff) (void)hipFree(L->d_cgoff);
  if (L->d_cg) (void)hipFree(L->d_cg);
  if (L->d_ncg) (void)hipFree(L->d_ncg);
  if (L->d_cpoff) (void)hipFree(L->d_cpoff);
  if (L->d_cp) (void)hipFree(L->d_cp);
  if (L->d_sfcode) (void)hipFree(L->d_sfcode);
  if (L->d_boxes) (void)hipFree(L->d_boxes);
  if (L->d_owner) (void)hipFree(L->d_owner);
  if (L->d_irr) (void)hipFree(L->d_irr);
  if (L->d_sfwg) (void)hipFree(L->d_sfwg);
  if (L->d_pfwg) (void)hipFree(L->d_pfwg);
  if (L->d_sfchunk) (void)hipFree(L->d_sfchunk);
  if (L->d_ring) (void)hipFree(L->d_ring);
  if (L->d_blist) (void)hipFree(L->d_blist);
  if (L->d_sfboxes) (void)hipFree(L->d_sfboxes);
  delete L;
}
extern "C" int pa_level_nboxes(const pa_level* L) { return L ? (int)L->boxes.size() : 0; }

// --------------------------------------------------------------------- MultiFab
extern "C" int64_t pa_mf_layout(int nboxes, const int32_t* b6, int ncomp, int ng, int64_t* off, int64_t* cstride) {
  int64_t t = 0;
  for (int b = 0; b < nboxes; ++b) {
    if (off) off[b] = t;
    int64_t n = 1;
    for (int d = 0; d < 3; ++d) n *= (int64_t)(b6[6 * b + 3 + d] - b6[6 * b + d] + 1 + 2 * ng);
    const int64_t cs = pa_cstride(n, ncomp);  // component stride (>= n, see pa_internal.h)
    if (cstride) cstride[b] = cs;
    t += cs * ncomp;  // every FAB (and every component) starts on a 512-byte boundary
  }
  return t;
}

extern "C" pa_mf* pa_mf_create(pa_ctx* ctx, const pa_level* L, int ncomp, int ng, double* devptr) {
  PaBind bind_(ctx);
  if (!ctx || !L) return nullptr;
  if (ncomp <= 0 || ng < 0) { pa_fail(ctx, "pa_mf_create: bad ncomp/ng"); return nullptr; }
  pa_mf* M = new pa_mf();
  M->lev = L; M->ncomp = ncomp; M->ng = ng;
  const int nb = (int)L->boxes.size();
  std::vector<int32_t> b6(6 * (size_t)nb);
  for (int b = 0; b < nb; ++b)
    for (int d = 0; d < 3; ++d) { b6[6 * b + d] = L->boxes[b].lo[d]; b6[6 * b + 3 + d] = L->boxes[b].hi[d]; }
  std::vector<int64_t> off(nb);
  M->total = pa_mf_layout(nb, b6.data(), ncomp, ng, off.data(), nullptr);
  M->off.assign(off.begin(), off.end());
  if (hipMalloc(&M->d_off, sizeof(long long) * std::max(nb, 1)) != hipSuccess ||
      (nb > 0 && hipMemcpy(M->d_off, M->off.data(), sizeof(long long) * nb, hipMemcpyHostToDevice) != hipSuccess)) {
    pa_fail(ctx, "pa_mf_create: device allocation failed");
    delete M;
    return nullptr;
  }
  if (devptr) {
    M->data = devptr;
  } else {
    if (hipMalloc(&M->data, sizeof(double) * (size_t)std::max<long long>(M->total, 1)) != hipSuccess ||
        hipMemsetAsync(M->data, 0, sizeof(double) * (size_t)M->total, ctx->stream) != hipSuccess) {
      pa_fail(ctx, "pa_mf_create: out of device memory (" + std::to_string(M->total * 8) + " bytes)");
      (void)hipFree(M->d_off);
      delete M;
      return nullptr;
    }
    M->owned = true;
  }
  M->view.data = M->data; M->view.off = M->d_off; M->view.ncomp = ncomp; M->view.ng = ng; M->view.xform = 0; M->view.xa = 0; M->view.xb = 1;
  return M;
}

extern "C" void pa_mf_destroy(pa_mf* M) {
  if (!M) return;
  if (M->owned && M->data) (void)hipFree(M->data);
  if (M->d_off) (void)hipFree(M->d_off);
  delete M;
}
extern "C" double* pa_mf_data(pa_mf* M) { return M ? M->data : nullptr; }
extern "C" int64_t pa_mf_size(const pa_mf* M) { return M ? M->total : 0; }

extern "C" int pa_mf_upload(pa_ctx* ctx, pa_mf* M, const double* host) {
  PaBind bind_(ctx);
  if (ctx && M && M->total == 0) return 0;  // a rank that owns no box of the level: nothing to move (host may be null)
  if (!ctx || !M || !host) return pa_fail(ctx, "pa_mf_upload: null argument");
  PA_HIP(hipMemcpyAsync(M->data, host, sizeof(double) * (size_t)M->total, hipMemcpyHostToDevice, ctx->stream));
  PA_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}
extern "C" int pa_mf_download(pa_ctx* ctx, const pa_mf* M, double* host) {
  PaBind bind_(ctx);
  if (ctx && M && M->total == 0) return 0;
  if (!ctx || !M || !host) return pa_fail(ctx, "pa_mf_download: null argument");
  PA_HIP(hipMemcpyAsync(host, M->data, sizeof(double) * (size_t)M->total, hipMemcpyDeviceToHost, ctx->stream));
  PA_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// components comp .. comp + ncomp - 1 only: inside a FAB a range of components is one contiguous piece ([comp][k][j][i]), so this
// is one copy per box.  host is laid out like the WHOLE multifab (pa_mf_layout); the other components of it are not touched.
// (grad3d uploaded its 5-component multifab whole to hand over one input component: 16 GB over the host link for 3.2 GB.)
static int mf_copy_comps(pa_ctx* ctx, const pa_mf* M, double* host, int comp, int ncomp, bool up, const char* who) {
  if (ctx && M && M->total == 0) return 0;
  if (!ctx || !M || !host) return pa_fail(ctx, std::string(who) + ": null argument");
  if (comp < 0 || ncomp < 1 || comp + ncomp > M->ncomp) return pa_fail(ctx, std::string(who) + ": component range");
  const pa_level* L = M->lev;
  for (size_t b = 0; b < L->boxes.size(); ++b) {
    const DBox& B = L->boxes[b];
    const long long cs = pa_cstride((long long)(B.hi[0] - B.lo[0] + 1 + 2 * M->ng) * (B.hi[1] - B.lo[1] + 1 + 2 * M->ng) * (B.hi[2] - B.lo[2] + 1 + 2 * M->ng), M->ncomp);
    const long long o = M->off[b] + (long long)comp * cs;
    if (up) PA_HIP(hipMemcpyAsync(M->data + o, host + o, sizeof(double) * (size_t)(cs * ncomp), hipMemcpyHostToDevice, ctx->stream));
    else PA_HIP(hipMemcpyAsync(host + o, M->data + o, sizeof(double) * (size_t)(cs * ncomp), hipMemcpyDeviceToHost, ctx->stream));
  }
  PA_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}
extern "C" int pa_mf_upload_comps(pa_ctx* ctx, pa_mf* M, const double* host, int comp, int ncomp) {
  PaBind bind_(ctx);
  return mf_copy_comps(ctx, M, const_cast<double*>(host), comp, ncomp, true, "pa_mf_upload_comps");
}
extern "C" int pa_mf_download_comps(pa_ctx* ctx, const pa_mf* M, double* host, int comp, int ncomp) {
  PaBind bind_(ctx);
  return mf_copy_comps(ctx, M, host, comp, ncomp, false, "pa_mf_download_comps");
}

__global__ void k_setval(double* p, long long n, double v) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

// per-box component ranges are contiguous: [comp][k][j][i]
__global__ void k_setval_comp(DLevelView L, DMFView M, int comp, int ncomp, double v) {
  const int b = blockIdx.y;
  const DBox B = L.boxes[b];
  const long long per = pa_cstride((long long)(B.hi[0] - B.lo[0] + 1 + 2 * M.ng) * (B.hi[1] - B.lo[1] + 1 + 2 * M.ng) *
                                       (B.hi[2] - B.lo[2] + 1 + 2 * M.ng), M.ncomp);
  double* p = M.data + M.off[b] + per * comp;
  const long long n = per * ncomp;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

extern "C" int pa_mf_setval(pa_ctx* ctx, pa_mf* M, int comp, int ncomp, double v) {
  PaBind bind_(ctx);
  if (!ctx || !M) return pa_fail(ctx, "pa_mf_setval: null argument");
  if (comp < 0 || comp + ncomp > M->ncomp) return pa_fail(ctx, "pa_mf_setval: component range");
  if (M->lev->boxes.empty()) return 0;  // a rank that owns no box of this level
  dim3 grid(64, (unsigned)M->lev->boxes.size());
  hipLaunchKernelGGL(k_setval_comp, grid, dim3(256), 0, ctx->stream, M->lev->view, M->view, comp, ncomp, v);
  PA_HIP(hipGetLastError());
  return 0;
}

__global__ void k_copy(DLevelView L, DMFView S, int scomp, DMFView D, int dcomp, int ncomp, int ng) {
  const int b = blockIdx.y;
  const DBox B = L.boxes[b];
  const int nx = B.hi[0] - B.lo[0] + 1 + 2 * ng, ny = B.hi[1] - B.lo[1] + 1 + 2 * ng, nz = B.hi[2] - B.lo[2] + 1 + 2 * ng;
  const long long n = (long long)nx * ny * nz * ncomp;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t % nx);
    const int j = (int)((t / nx) % ny);
    const int k = (int)((t / ((long long)nx * ny)) % nz);
    const int c = (int)(t / ((long long)nx * ny * nz));
    const int I = B.lo[0] - ng + i, J = B.lo[1] - ng + j, K = B.lo[2] - ng + k;
    D.data[D.off[b] + fab_index(B, D.ng, D.ncomp, dcomp + c, I, J, K)] = S.data[S.off[b] + fab_index(B, S.ng, S.ncomp, scomp + c, I, J, K)];
  }
}

extern "C" int pa_mf_copy(pa_ctx* ctx, const pa_mf* S, int scomp, pa_mf* D, int dcomp, int ncomp, int ng) {
  PaBind bind_(ctx);
  if (!ctx || !S || !D) return pa_fail(ctx, "pa_mf_copy: null argument");
  if (S->lev != D->lev) return pa_fail(ctx, "pa_mf_copy: different levels");
  if (ng > S->ng || ng > D->ng || scomp + ncomp > S->ncomp || dcomp + ncomp > D->ncomp)
    return pa_fail(ctx, "pa_mf_copy: ng/component range");
  if (S->lev->boxes.empty()) return 0;  // a rank that owns no box of this level
  dim3 grid(128, (unsigned)S->lev->boxes.size());
  hipLaunchKernelGGL(k_copy, grid, dim3(256), 0, ctx->stream, S->lev->view, S->view, scomp, D->view, dcomp, ncomp, ng);
  PA_HIP(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------- FillBoundary
// Thread per ghost-shell cell.  The shell of depth ng is enumerated as 2 z-slabs (full grown
// xy extent), 2 y-slabs (valid z, full grown x) and 2 x-slabs (valid y,z): x-contiguous runs.
// (32-bit unsigned index arithmetic: a 64-bit division costs hundreds of instructions on the GPU;
// the shell of one box is far below 2^32 cells)
__device__ __forceinline__ bool shell_cell(const DBox& B, int ng, long long tt, int& i, int& j, int& k) {
  const unsigned nx = B.hi[0] - B.lo[0] + 1, ny = B.hi[1] - B.lo[1] + 1, nz = B.hi[2] - B.lo[2] + 1;
  const unsigned g = (unsigned)ng, gx = nx + 2 * g, gy = ny + 2 * g;
  const unsigned nzs = g * gy * gx;  // one z slab
  const unsigned nys = nz * g * gx;  // one y slab
  const unsigned nxs = nz * ny * g;  // one x slab
  if (tt >= 2LL * nzs + 2LL * nys + 2LL * nxs) return false;
  unsigned t = (unsigned)tt;
  if (t < 2 * nzs) {
    const int s = t >= nzs;
    if (s) t -= nzs;
    const unsigned r = t / gx;
    i = B.lo[0] - ng + (int)(t - r * gx);
    const unsigned kk = r / gy;
    j = B.lo[1] - ng + (int)(r - kk * gy);
    k = s ? B.hi[2] + 1 + (int)kk : B.lo[2] - ng + (int)kk;
    return true;
  }
  t -= 2 * nzs;
  if (t < 2 * nys) {
    const int s = t >= nys;
    if (s) t -= nys;
    const unsigned r = t / gx;
    i = B.lo[0] - ng + (int)(t - r * gx);
    const unsigned kk = r / g;
    const int jj = (int)(r - kk * g);
    k = B.lo[2] + (int)kk;
    j = s ? B.hi[1] + 1 + jj : B.lo[1] - ng + jj;
    return true;
  }
  t -= 2 * nys;
  {
    const int s = t >= nxs;
    if (s) t -= nxs;
    const unsigned r = t / g;
    const int ii = (int)(t - r * g);
    const unsigned kk = r / ny;
    j = B.lo[1] + (int)(r - kk * ny);
    k = B.lo[2] + (int)kk;
    i = s ? B.hi[0] + 1 + ii : B.lo[0] - ng + ii;
    return true;
  }
}

struct FillArgs { DLevelView L; DMFView M; int comp, ncomp, ngf; };
__global__ void k_fill_boundary(LevBatch<FillArgs> Bt) {
  unsigned yb;
  const FillArgs& Fa = Bt.a[Bt.find(blockIdx.y, yb)];
  const DLevelView& L = Fa.L;
  const DMFView& M = Fa.M;
  const int comp = Fa.comp, ncomp = Fa.ncomp, ngf = Fa.ngf;
  const int b = (int)yb;
  const DBox B = L.boxes[b];
  const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  int i, j, k;
  if (!shell_cell(B, ngf, t, i, j, k)) return;
  int s, p[3];
  if (classify(L, i, j, k, s, p) != 0 || s < 0) return;  // s < 0: covered by another rank's box (filled by the exchange)
  const DBox S = L.boxes[s];
  for (int c = comp; c < comp + ncomp; ++c)
    M.data[M.off[b] + fab_index(B, M.ng, M.ncomp, c, i, j, k)] = M.data[M.off[s] + fab_index(S, M.ng, M.ncomp, c, p[0], p[1], p[2])];
}

// the same copies from the level's region plan (pa_dist.h: FbLocal): blockIdx.y = level of the batch, blockIdx.x = its workgroup table
struct FbrArgs { DLevelView L; DMFView M; const int* regs; const int* wgs; int nwg, comp, ncomp; };
__global__ __launch_bounds__(256) void k_fill_boundary_regions(LevBatch<FbrArgs> Bt) {
  const FbrArgs& Fa = Bt.a[blockIdx.y];
  if ((int)blockIdx.x >= Fa.nwg) return;
  const int* R = Fa.regs + 16 * Fa.wgs[2 * blockIdx.x];
  const unsigned t = (unsigned)Fa.wgs[2 * blockIdx.x + 1] * 256u + threadIdx.x;
  if (t >= (unsigned)R[11]) return;
  const DMFView& M = Fa.M;
  const DBox D = Fa.L.boxes[R[0]], S = Fa.L.boxes[R[1]];
  const int ng = M.ng;
  const unsigned r = R[5] == 1 ? t : __umulhi(t, (unsigned)R[12]), i = t - r * (unsigned)R[5];  // (ceil(2^32 / 1) does not fit)
  const unsigned k = R[6] == 1 ? r : __umulhi(r, (unsigned)R[13]), j = r - k * (unsigned)R[6];
  const int nxd = D.hi[0] - D.lo[0] + 1 + 2 * ng, nyd = D.hi[1] - D.lo[1] + 1 + 2 * ng, nzd = D.hi[2] - D.lo[2] + 1 + 2 * ng;
  const int nxs = S.hi[0] - S.lo[0] + 1 + 2 * ng, nys = S.hi[1] - S.lo[1] + 1 + 2 * ng, nzs = S.hi[2] - S.lo[2] + 1 + 2 * ng;
  const long long csd = pa_cstride((long long)nxd * nyd * nzd, M.ncomp), css = pa_cstride((long long)nxs * nys * nzs, M.ncomp);
  // region origin inside the two FABs (wave-uniform), then the cell
  const int dk = R[4] - D.lo[2] + ng, dj = R[3] - D.lo[1] + ng, di = R[2] - D.lo[0] + ng;
  const int sk = R[4] - R[10] - S.lo[2] + ng, sj = R[3] - R[9] - S.lo[1] + ng, si = R[2] - R[8] - S.lo[0] + ng;
  double* dst = M.data + M.off[R[0]] + (long long)Fa.comp * csd + ((long long)(dk + (int)k) * nyd + (dj + (int)j)) * nxd + (di + (int)i);
  const double* src = M.data + M.off[R[1]] + (long long)Fa.comp * css + ((long long)(sk + (int)k) * nys + (sj + (int)j)) * nxs + (si + (int)i);
  if (gridDim.z > 1) {  // components as a grid dimension: a wave stays inside one component's pages
    dst[blockIdx.z * csd] = src[blockIdx.z * css];
    return;
  }
  for (int c = 0; c < Fa.ncomp; ++c) dst[c * csd] = src[c * css];
}

int pa_fill_boundary_local_batch(pa_ctx* ctx, int n, pa_mf* const* Ms, int comp, int ncomp, int ng);
int pa_fill_boundary_local_batch_ngs(pa_ctx* ctx, int n, pa_mf* const* Ms, int comp, int ncomp, const int* ngs);
static long long max_shell(const pa_level* L, int ng) {
  long long m = 0;
  for (const DBox& B : L->boxes) {
    const long long nx = B.hi[0] - B.lo[0] + 1, ny = B.hi[1] - B.lo[1] + 1, nz = B.hi[2] - B.lo[2] + 1;
    m = std::max(m, (nx + 2 * ng) * (ny + 2 * ng) * (nz + 2 * ng) - nx * ny * nz);
  }
  return m;
}

// the local half of FillBoundary on several levels (same component range and ghost width) in as few launches as possible
int pa_fill_boundary_local_batch(pa_ctx* ctx, int n, pa_mf* const* Ms, int comp, int ncomp, int ng) {
  std::vector<int> ngs((size_t)std::max(n, 1), ng);
  return pa_fill_boundary_local_batch_ngs(ctx, n, Ms, comp, ncomp, ngs.data());
}
// ... with a ghost width per level (filterPlt's levels: ngrow 1 / 2 / 4; pa_fill_ghosts_hierarchy); ngs[i] = 0: level i is skipped
int pa_fill_boundary_local_batch_ngs(pa_ctx* ctx, int n, pa_mf* const* Ms, int comp, int ncomp, const int* ngs) {
  {  // copy regions when every level of the batch has a plan (pa_dist.hip)
    bool regions = true;
    std::vector<FbLocal*> plans(n, nullptr);
    for (int i = 0; i < n && regions; ++i) {
      if (Ms[i]->lev->boxes.empty() || ngs[i] <= 0) continue;
      plans[i] = pa_fb_local_plan(ctx, Ms[i]->lev, ngs[i]);
      regions = plans[i] && plans[i]->ok;
    }
    if (regions) {
      for (int i0 = 0; i0 < n; i0 += PA_MAXB) {
        LevBatch<FbrArgs> Bt;
        int mw = 0;
        for (int i = i0; i < n && i < i0 + PA_MAXB; ++i) {
          if (!plans[i] || plans[i]->nwg == 0) continue;
          Bt.a[Bt.n] = FbrArgs{Ms[i]->lev->view, Ms[i]->view, plans[i]->d_regs, plans[i]->d_wgs, plans[i]->nwg, comp, ncomp};
          ++Bt.n;
          mw = std::max(mw, plans[i]->nwg);
        }
        if (!Bt.n) continue;
        // several components: the component is a grid dimension, not a loop inside the thread -- a wave then stays inside one
        // component's pages (config 2's 10 components: 1.27 -> 1.03-1.10 ms, config 5's shape 1.58 -> 1.34-1.38) than a loop over the components
        const unsigned gz = (ncomp > 1 && ncomp <= 65535) ? (unsigned)ncomp : 1u;
        hipLaunchKernelGGL(k_fill_boundary_regions, dim3((unsigned)mw, (unsigned)Bt.n, gz), dim3(256), 0, ctx->stream, Bt);
      }
      PA_HIP(hipGetLastError());
      return 0;
    }
  }
  for (int i0 = 0; i0 < n; i0 += PA_MAXB) {
    LevBatch<FillArgs> Bt;
    long long ms = 0;
    for (int i = i0; i < n && i < i0 + PA_MAXB; ++i) {
      if (Ms[i]->lev->boxes.empty() || ngs[i] <= 0) continue;
      Bt.a[Bt.n] = FillArgs{Ms[i]->lev->view, Ms[i]->view, comp, ncomp, ngs[i]};
      Bt.ycum[Bt.n + 1] = Bt.ycum[Bt.n] + (int)Ms[i]->lev->boxes.size();
      ++Bt.n;
      ms = std::max(ms, max_shell(Ms[i]->lev, ngs[i]));
    }
    if (!Bt.n) continue;
    hipLaunchKernelGGL(k_fill_boundary, dim3((unsigned)((ms + 255) / 256), (unsigned)Bt.ycum[Bt.n]), dim3(256), 0, ctx->stream, Bt);
  }
  PA_HIP(hipGetLastError());
  return 0;
}

// no_exchange: only the local half (the caller batches the cross-rank half of several levels into one exchange)
int pa_fill_boundary_impl(pa_ctx* ctx, pa_mf* M, int comp, int ncomp, int ng, int no_exchange) {
  if (!ctx || !M) return pa_fail(ctx, "pa_fill_boundary: null argument");
  if (ng > M->ng || ng < 0 || comp < 0 || comp + ncomp > M->ncomp) return pa_fail(ctx, "pa_fill_boundary: ng/component range");
  if (ng == 0) return 0;
  for (int d = 0; d < 3; ++d)
    if (M->lev->is_per[d] && ng > M->lev->domhi[d] - M->lev->domlo[d] + 1)
      return pa_fail(ctx, "pa_fill_boundary: ng larger than the periodic domain");
  ProfScope prof(ctx, PA_TAG_FILL);
  if (!M->lev->boxes.empty()) {
    pa_mf* one[1] = {M};
    if (pa_fill_boundary_local_batch(ctx, 1, one, comp, ncomp, ng)) return 1;
  }
  // ghost cells covered by boxes of other ranks: the cross-rank half (pack -> grouped send/recv -> unpack)
  if (M->lev->nranks > 1 && !no_exchange) {
    XPlan* P = pa_fb_plan(ctx, M->lev, ng);
    if (!P) return 1;
    XJob J = {P, M, comp, M, comp, ncomp};
    if (pa_xexchange(ctx, 1, &J)) return 1;
  }
  return 0;
}
extern "C" int pa_fill_boundary(pa_ctx* ctx, pa_mf* M, int comp, int ncomp, int ng) {
  PaBind bind_(ctx);
  return pa_fill_boundary_impl(ctx, M, comp, ncomp, ng, 0);
}

// --------------------------------------------------------------------- applyBC
struct BCArgs {
  int bc[3];
  int ratio;
  int only_dir;
  int has_crse;
  int edges;  // 0: face ghosts (AMReX applyBC); 1: the 12 edge-ghost lines (fused path extension)
};

// resolved boundary value of ghost cell q of box B (fab pointer f) whose interior neighbour
// in direction dir is q + s*e_dir.  cls: 1 coarse-fine, 2 physical wall.
__device__ __forceinline__ double bc_ghost_value(const DLevelView& L, const DMFView& M, const DBox& B, int b, int comp,
                                                 const DLevelView& LC, const DMFView& MC, int ccomp, const BCArgs& A,
                                                 const int q[3], int dir, int s, int cls, bool& ok) {
  int in[3] = {q[0], q[1], q[2]};
  in[dir] += s;
  const double* f = M.data + M.off[b];
  if (cls == 2) {
    const double v = f[fab_index(B, M.ng, M.ncomp, comp, in[0], in[1], in[2])];
    return (A.bc[dir] == PA_BC_REFLECT_ODD) ? -v : v;
  }
  double coef[4];
  const int NX = cf_normal_coef(B.hi[dir] - B.lo[dir] + 1, A.ratio, coef);
  const double bv = cf_bndry_value(L, LC, MC, ccomp, q, dir, A.ratio, ok);
  double tmp = 0.0;
  for (int m = 1; m < NX; ++m) {
    int pc[3] = {q[0], q[1], q[2]};
    pc[dir] += s * m;
    tmp += f[fab_index(B, M.ng, M.ncomp, comp, pc[0], pc[1], pc[2])] * coef[m];
  }
  double g = tmp;
  g += bv * coef[0];
  return g;
}

// MLMG applyBC on the face ghost cells of NF fields of one level at once (same ghost cells, same
// masks and weights; the coarse-fine boundary values of all fields come from ONE coarse component,
// see cf_bndry_values).  Thread per ghost cell of a special face.
template <int NF>
struct BCFields {
  DMFView M[NF];
  int comp[NF];
  int xf[NF];
};
// wg != null: workgroup w works on chunk wg[w].y (256 ghost cells) of special face wg[w].x (the level's work table, pa_level::d_sfwg:
// faces differ 16x in size on general BoxArrays, a grid of (largest face / 256) x faces is mostly empty workgroups there); else
// blockIdx.y = special face, blockIdx.x = chunk
template <int NF>
__global__ __launch_bounds__(256) void k_apply_bc_sfaces(DLevelView L, BCFields<NF> F, DLevelView LC, DMFView MC, int ccomp, BCArgs A, int* nbad, const int2* wg = nullptr) {
  int b, dir, side, layer, q[3];
  DBox B;
  const unsigned fy = wg ? (unsigned)wg[blockIdx.x].x : blockIdx.y;
  const long long t = (wg ? (long long)wg[blockIdx.x].y : (long long)blockIdx.x) * (long long)blockDim.x + threadIdx.x;
  if (!sface_decode(L, fy, t, 1, b, B, dir, side, q, layer)) return;
  if (A.only_dir >= 0 && dir != A.only_dir) return;
  const unsigned code = L.sfcode[L.sfoff[fy] + t];
  const int cls = (int)(code & 3u);
  if (cls == 0) return;
  if (cls == 1 && !A.has_crse) { atomicAdd(nbad, 1); return; }
  const int s = side ? -1 : 1;
  if (cls == 2) {
    int in[3] = {q[0], q[1], q[2]};
    in[dir] += s;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      double* p = F.M[f].data + F.M[f].off[b];
      const double v = p[fab_index(B, F.M[f].ng, F.M[f].ncomp, F.comp[f], in[0], in[1], in[2])];
      p[fab_index(B, F.M[f].ng, F.M[f].ncomp, F.comp[f], q[0], q[1], q[2])] = (A.bc[dir] == PA_BC_REFLECT_ODD) ? -v : v;
    }
    return;
  }
  bool ok = true;
  double coef[4], bv[NF];
  const int NX = cf_normal_coef(B.hi[dir] - B.lo[dir] + 1, A.ratio, coef);
  cf_interp<NF>(code, LC, MC, ccomp, q, dir, A.ratio, F.xf, ok, bv);
  if (!ok) atomicAdd(nbad, 1);
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    double* p = F.M[f].data + F.M[f].off[b];
    double tmp = 0.0;
    for (int m = 1; m < NX; ++m) {
      int pc[3] = {q[0], q[1], q[2]};
      pc[dir] += s * m;
      tmp += p[fab_index(B, F.M[f].ng, F.M[f].ncomp, F.comp[f], pc[0], pc[1], pc[2])] * coef[m];
    }
    double g = tmp;
    g += bv[f] * coef[0];
    p[fab_index(B, F.M[f].ng, F.M[f].ncomp, F.comp[f], q[0], q[1], q[2])] = g;
  }
}

// Fused-path extension: edge ghost cells (outside the box in two directions a<c).  Such a cell
// is needed as the boundary ghost of a valid cell of a NEIGHBOURING box (role = the direction
// towards that valid cell); see DESIGN.md "resolved ghost ring".
__global__ void k_apply_bc_edges(DLevelView L, DMFView M, int comp, DLevelView LC, DMFView MC, int ccomp, BCArgs A,
                                 int* nbad) {
  const int b = blockIdx.y;
  const DBox B = L.boxes[b];
  const int n[3] = {B.hi[0] - B.lo[0] + 1, B.hi[1] - B.lo[1] + 1, B.hi[2] - B.lo[2] + 1};
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  // 3 edge orientations e (the free direction), 4 edges each
  int e = -1, which = 0, pos = 0;
  for (int d = 0; d < 3; ++d) {
    if (t < 4LL * n[d]) { e = d; which = (int)((unsigned)t / (unsigned)n[d]); pos = (int)((unsigned)t % (unsigned)n[d]); break; }
    t -= 4LL * n[d];
  }
  if (e < 0) return;
  const int a = (e == 0) ? 1 : 0, c = (e == 2) ? 1 : 2;
  const int sa = which & 1, sc = which >> 1;
  int q[3];
  q[e] = B.lo[e] + pos;
  q[a] = sa ? B.hi[a] + 1 : B.lo[a] - 1;
  q[c] = sc ? B.hi[c] + 1 : B.lo[c] - 1;
  const int cls = classify(L, q[0], q[1], q[2]);
  if (cls == 0) return;
  int qa[3] = {q[0], q[1], q[2]}, qc[3] = {q[0], q[1], q[2]};
  qa[a] += sa ? -1 : 1;
  qc[c] += sc ? -1 : 1;
  const bool va = classify(L, qa[0], qa[1], qa[2]) == 0;
  const bool vc = classify(L, qc[0], qc[1], qc[2]) == 0;
  if (va == vc) return;  // neither: unused; both: non-fusable level (checked on the host)
  const int dir = va ? a : c;
  const int s = va ? (sa ? -1 : 1) : (sc ? -1 : 1);
  // class of q as a ghost in direction dir: outside the domain along dir -> wall
  int cl = cls;
  if (cls == 2) {
    const bool out_dir = (q[dir] < L.domlo[dir] || q[dir] > L.domhi[dir]) && !L.is_per[dir];
    if (!out_dir) return;  // outside along the other direction only: the neighbour is not valid either
  }
  if (cl == 1 && !A.has_crse) { atomicAdd(nbad, 1); return; }
  bool ok = true;
  const double v = bc_ghost_value(L, M, B, b, comp, LC, MC, ccomp, A, q, dir, s, cl, ok);
  if (!ok) atomicAdd(nbad, 1);
  M.data[M.off[b] + fab_index(B, M.ng, M.ncomp, comp, q[0], q[1], q[2])] = v;
}

int pa_ensure_red(pa_ctx* ctx, size_t n) {
  if (ctx->red_cap >= n) return 0;
  if (ctx->d_red) (void)hipFree(ctx->d_red);
  ctx->d_red = nullptr;
  ctx->red_cap = 0;
  PA_HIP(hipMalloc(&ctx->d_red, n * sizeof(double)));
  ctx->red_cap = n;
  return 0;
}

// number of coarse-fine ghost cells, since the last call, whose coarse data was missing
// (improper nesting or a level-0 box not covering the domain).  Synchronous; resets the count.
extern "C" int pa_bc_errors(pa_ctx* ctx) {
  PaBind bind_(ctx);
  if (!ctx) return -1;
  int n = 0;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipMemcpy(&n, ctx->d_flags, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemset(ctx->d_flags, 0, sizeof(int)) != hipSuccess)
    return -1;
  return n;
}

int pa_apply_bc_impl(pa_ctx* ctx, pa_mf* F, int comp, const pa_mf* C, int ccomp, const int32_t bc[3], int ratio,
                     int only_dir, int edges, const double* crse_xform) {
  if (!ctx || !F) return pa_fail(ctx, "pa_apply_bc: null argument");
  if (F->ng < 1) return pa_fail(ctx, "pa_apply_bc: multifab has no ghost cells");
  if (comp < 0 || comp >= F->ncomp || (C && (ccomp < 0 || ccomp >= C->ncomp))) return pa_fail(ctx, "pa_apply_bc: component range");
  if (ratio != 2 && C) return pa_fail(ctx, "pa_apply_bc: only refinement ratio 2 is supported (quirk Q11)");
  int* nbad = ctx->d_flags;
  const pa_level* L = F->lev;
  // sharded coarse level: this rank's coarse-source copy, refilled here (callers that batch the refill of several
  // levels pass the coarse-source multifab itself).  Null afterwards: no coarse-fine ghost cell on this rank.
  if (C && pa_coarse_source(ctx, L, C, ccomp, 1, 0, 0, 0, &C, &ccomp)) return 1;
  BCArgs A;
  for (int d = 0; d < 3; ++d) A.bc[d] = bc[d];
  A.ratio = ratio; A.only_dir = only_dir; A.has_crse = C ? 1 : 0; A.edges = edges;
  if (L->boxes.empty()) return 0;
  DLevelView LC = C ? C->lev->view : L->view;
  DMFView MC = C ? C->view : F->view;
  MC.xform = 0;
  if (crse_xform) { MC.xform = 1; MC.xa = crse_xform[0]; MC.xb = crse_xform[1]; }  // coarse value = (v - xa) * xb
  const long long n0 = L->maxn[0], n1 = L->maxn[1], n2 = L->maxn[2];
  ProfScope prof(ctx, PA_TAG_BC);
  if (!edges) {
    if (L->sfaces.empty()) return 0;  // every ghost cell is a valid cell of the level
    const long long nt = std::max(n1 * n2, std::max(n0 * n2, n0 * n1));
    dim3 grid((unsigned)((nt + 255) / 256), (unsigned)L->sfaces.size());
    if (L->d_sfwg) grid = dim3((unsigned)L->nsfwg);
    BCFields<1> Fs;
    Fs.M[0] = F->view; Fs.comp[0] = comp; Fs.xf[0] = MC.xform;
    hipLaunchKernelGGL(k_apply_bc_sfaces<1>, grid, dim3(256), 0, ctx->stream, L->view, Fs, LC, MC, ccomp, A, nbad, (const int2*)L->d_sfwg);
  } else {
    const long long nt = 4 * (n0 + n1 + n2);
    dim3 grid((unsigned)((nt + 255) / 256), (unsigned)L->boxes.size());
    hipLaunchKernelGGL(k_apply_bc_edges, grid, dim3(256), 0, ctx->stream, L->view, F->view, comp, LC, MC, ccomp, A, nbad);
  }
  PA_HIP(hipGetLastError());
  return 0;
}

// applyBC on the face ghosts of two fields of one level in one launch: F0/comp0 takes the coarse
// component as it is, F1/comp1 sees it through the affine view (v - xform[0]) * xform[1].
int pa_apply_bc_dual(pa_ctx* ctx, pa_mf* F0, int comp0, pa_mf* F1, int comp1, const pa_mf* C, int ccomp, const int32_t bc[3], int ratio,
                     const double* xform) {
  if (!ctx || !F0 || !F1 || !xform) return pa_fail(ctx, "pa_apply_bc_dual: null argument");
  if (F0->lev != F1->lev) return pa_fail(ctx, "pa_apply_bc_dual: different levels");
  if (F0->ng < 1 || F1->ng < 1) return pa_fail(ctx, "pa_apply_bc: multifab has no ghost cells");
  if (comp0 < 0 || comp0 >= F0->ncomp || comp1 < 0 || comp1 >= F1->ncomp || (C && (ccomp < 0 || ccomp >= C->ncomp)))
    return pa_fail(ctx, "pa_apply_bc: component range");
  if (ratio != 2 && C) return pa_fail(ctx, "pa_apply_bc: only refinement ratio 2 is supported (quirk Q11)");
  const pa_level* L = F0->lev;
  if (C && pa_coarse_source(ctx, L, C, ccomp, 1, 0, 0, 0, &C, &ccomp)) return 1;
  if (L->sfaces.empty()) return 0;
  BCArgs A;
  for (int d = 0; d < 3; ++d) A.bc[d] = bc[d];
  A.ratio = ratio; A.only_dir = -1; A.has_crse = C ? 1 : 0; A.edges = 0;
  DLevelView LC = C ? C->lev->view : L->view;
  DMFView MC = C ? C->view : F0->view;
  MC.xform = 1; MC.xa = xform[0]; MC.xb = xform[1];
  BCFields<2> Fs;
  Fs.M[0] = F0->view; Fs.comp[0] = comp0; Fs.xf[0] = 0;
  Fs.M[1] = F1->view; Fs.comp[1] = comp1; Fs.xf[1] = 1;
  const long long n0 = L->maxn[0], n1 = L->maxn[1], n2 = L->maxn[2];
  const long long nt = std::max(n1 * n2, std::max(n0 * n2, n0 * n1));
  dim3 grid((unsigned)((nt + 255) / 256), (unsigned)L->sfaces.size());
  if (L->d_sfwg) grid = dim3((unsigned)L->nsfwg);
  ProfScope prof(ctx, PA_TAG_BC);
  hipLaunchKernelGGL(k_apply_bc_sfaces<2>, grid, dim3(256), 0, ctx->stream, L->view, Fs, LC, MC, ccomp, A, ctx->d_flags, (const int2*)L->d_sfwg);
  PA_HIP(hipGetLastError());
  return 0;
}

extern "C" int pa_apply_bc(pa_ctx* ctx, pa_mf* F, int comp, const pa_mf* C, int ccomp, const int32_t bc[3], int ratio,
                           int only_dir) {
  PaBind bind_(ctx);
  return pa_apply_bc_impl(ctx, F, comp, C, ccomp, bc, ratio, only_dir, 0, nullptr);
}

// ------------------------------------------------------------ progress variable, shell only
// c = (s - pmin) * invdenom (curvature.cpp:316-320) stored only where the fused path reads it: for
// every special face of a box, the slab of `depth` valid layers + ng ghost layers along the face
// normal, over the full grown tangential extent.  Slabs of one box overlap at its edges (same value).
__global__ __launch_bounds__(256) void k_progress_shell(DLevelView L, DMFView S, int comp, DMFView Cm, int ccomp, int ng, int depth, double pmin, double invdenom) {
  const int e = L.sfaces[blockIdx.y];
  const int b = e / 6, dir = (e % 6) >> 1, side = e & 1;
  const DBox B = L.boxes[b];
  int o[3], n[3];
  for (int d = 0; d < 3; ++d) { o[d] = B.lo[d] - ng; n[d] = B.hi[d] - B.lo[d] + 1 + 2 * ng; }
  n[dir] = ng + depth;
  if (side) o[dir] = B.hi[dir] - depth + 1;
  const long long tt = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (tt >= (long long)n[0] * n[1] * n[2]) return;
  const unsigned t = (unsigned)tt, r = t / (unsigned)n[0], kk = r / (unsigned)n[1];  // 32-bit: see shell_cell
  const int i = o[0] + (int)(t - r * (unsigned)n[0]), j = o[1] + (int)(r - kk * (unsigned)n[1]), k = o[2] + (int)kk;
  Cm.data[Cm.off[b] + fab_index(B, Cm.ng, Cm.ncomp, ccomp, i, j, k)] = (S.data[S.off[b] + fab_index(B, S.ng, S.ncomp, comp, i, j, k)] - pmin) * invdenom;
}

extern "C" int pa_progress_shell_level(pa_ctx* ctx, const pa_mf* s, int comp, double pmin, double pmax, pa_mf* c, int ccomp, int ng, int depth) {
  PaBind bind_(ctx);
  if (!ctx || !s || !c) return pa_fail(ctx, "pa_progress_shell_level: null argument");
  if (s->lev != c->lev) return pa_fail(ctx, "pa_progress_shell_level: different levels");
  if (ng > s->ng || ng > c->ng || comp >= s->ncomp || ccomp >= c->ncomp || depth < 1) return pa_fail(ctx, "pa_progress_shell_level: ng/component range");
  const pa_level* L = s->lev;
  for (const DBox& B : L->boxes)
    for (int d = 0; d < 3; ++d)
      if (B.hi[d] - B.lo[d] + 1 <= 2 * depth)  // no interior core left: the shell is the whole box
        return pa_progress_level(ctx, s, comp, pmin, pmax, c, ccomp, ng);
  if (L->sfaces.empty()) return 0;
  if (s->lev->boxes.empty()) return 0;  // a rank that owns no box of this level
  const long long g0 = L->maxn[0] + 2 * ng, g1 = L->maxn[1] + 2 * ng, g2 = L->maxn[2] + 2 * ng;
  const long long ms = (long long)(ng + depth) * std::max(g1 * g2, std::max(g0 * g2, g0 * g1));
  dim3 grid((unsigned)((ms + 255) / 256), (unsigned)L->sfaces.size());
  ProfScope prof(ctx, PA_TAG_PROGRESS);
  hipLaunchKernelGGL(k_progress_shell, grid, dim3(256), 0, ctx->stream, L->view, s->view, comp, c->view, ccomp, ng, depth, pmin, 1.0 / (pmax - pmin));
  PA_HIP(hipGetLastError());
  return 0;
}

extern "C" void* pa_device_malloc(pa_ctx* ctx, int64_t bytes) {
  PaBind bind_(ctx);
  if (!ctx || bytes < 0) return nullptr;
  void* p = nullptr;
  if (hipMalloc(&p, (size_t)(bytes > 0 ? bytes : 8)) != hipSuccess) { pa_fail(ctx, "pa_device_malloc: out of device memory"); return nullptr; }
  return p;
}
extern "C" void pa_device_free(pa_ctx* ctx, void* p) {
  PaBind bind_(ctx);
  if (!p) return;
  if (ctx) {
    auto it = ctx->surf_live.find(p);
    if (it != ctx->surf_live.end()) {  // a surface of pa_mc_level*: keep the block for the next one
      const size_t bytes = it->second;
      ctx->surf_live.erase(it);
      if (ctx->surf_cache.size() < 4) { ctx->surf_cache.emplace_back(p, bytes); return; }
    }
  }
  (void)hipFree(p);
}
extern "C" int pa_memcpy_h2d(pa_ctx* ctx, void* dst, const void* src, int64_t bytes) {
  PaBind bind_(ctx);
  if (!ctx || (bytes > 0 && (!dst || !src))) return pa_fail(ctx, "pa_memcpy_h2d: null argument");
  PA_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
  PA_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}
extern "C" int pa_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
extern "C" int pa_memcpy_d2d(pa_ctx* ctx, void* dst, const void* src, int64_t bytes) {
  PaBind bind_(ctx);
  if (!ctx || (bytes > 0 && (!dst || !src))) return pa_fail(ctx, "pa_memcpy_d2d: null argument");
  PA_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDefault, ctx->stream));  // unified addressing: either device
  PA_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}
extern "C" int pa_memcpy_d2h(pa_ctx* ctx, void* dst, const void* src, int64_t bytes) {
  PaBind bind_(ctx);
  if (!ctx || (bytes > 0 && (!dst || !src))) return pa_fail(ctx, "pa_memcpy_d2h: null argument");
  PA_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
  PA_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

