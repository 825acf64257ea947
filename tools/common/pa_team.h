// pa_team.h -- ngpus=<n> for the tool drivers: n ranks = n host threads of ONE process, one pa_ctx (one GPU) each.
//
// The reference's tools are MPI programs: every rank owns the FABs DistributionMapping(ba) gives it (grad.cpp:162,
// curvature.cpp:289) and AMReX moves ghost data between ranks.  Here the plotfile is read once, the boxes of every level
// are dealt to the ranks by pa_distribution_map, every rank creates its share with pa_level_create_sharded and runs the
// SAME library pipeline as a 1-GPU run -- the library does the cross-rank ghost fills through the rank's transport:
//   * RCCL over xGMI (pa_ctx_init_rccl: one communicator rank per thread) when every rank has its own GPU;
//   * an in-process transport (pairwise peer copies between the contexts' buffers through FIFO mailboxes) when ranks share a GPU
//     (gpu_share=1 / fewer GPUs than ranks: how the tests run n > 1 on a one-GPU box) or RCCL is unavailable.
// Results are bit-identical for every n, so the output files are byte-identical.
#pragma once
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "pa_device.h"

namespace pa {

class Barrier {
  std::mutex m;
  std::condition_variable cv;
  int n, count = 0;
  unsigned gen = 0;

 public:
  explicit Barrier(int n_) : n(n_) {}
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const unsigned g = gen;
    if (++count == n) { count = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
};

struct Team {
  int n = 1;
  bool share = false;
  std::string transport = "none";
  std::vector<std::unique_ptr<Ctx>> ctx;
  struct User { Team* t; int r; };
  std::vector<User> users;
  std::vector<pa_comm> comms;
  std::unique_ptr<Barrier> bar;
  std::vector<std::vector<double>> red;

  explicit Team(const ParmParse& pp) {
    pp.query("ngpus", n);
    if (n < 1) Abort("ngpus must be >= 1");
    int sh = -1;
    pp.query("gpu_share", sh);
    const int ndev = pa_device_count();
    if (ndev < 1) Abort("no MI355X / HIP device available (this build has no CPU fallback)");
    share = sh >= 0 ? sh != 0 : ndev < n;
    if (!share && ndev < n) Abort("ngpus exceeds the number of visible GPUs (gpu_share=1 lets ranks share them)");
    ctx.resize(n);
    users.resize(n);
    comms.resize(n);
    box.resize((size_t)n * n);
    red.resize(n);
    bar.reset(new Barrier(n));
    for (int r = 0; r < n; ++r) users[r] = User{this, r};
    if (n == 1) { ctx[0].reset(new Ctx(0)); return; }
    // contexts + transport, one thread per rank (ncclCommInitRank blocks until every rank has called it)
    std::vector<char> id(128, 0);
    std::vector<int> rccl_ok(n, 0);
    bool try_rccl = !share;
    ctx[0].reset(new Ctx(0));
    if (try_rccl && pa_rccl_unique_id(ctx[0]->h, id.data()) != 0) try_rccl = false;
    run_threads([&](int r) {
      if (r > 0) ctx[r].reset(new Ctx(share ? r % ndev : r));
      if (try_rccl) rccl_ok[r] = pa_ctx_init_rccl(ctx[r]->h, n, r, id.data()) == 0 && pa_comm_selftest(ctx[r]->h, 1 << 14) == 0;
    });
    bool all = try_rccl;
    for (int r = 0; r < n; ++r) all = all && rccl_ok[r];
    if (all) { transport = "RCCL (grouped ncclSend/ncclRecv per rank thread)"; return; }
    transport = share ? "in-process peer copies (ranks share GPUs)" : "in-process peer copies (RCCL unavailable)";
    for (int r = 0; r < n; ++r) {
      comms[r] = pa_comm{&users[r], r, n, &Team::exchange, &Team::allreduce};
      ctx[r]->check(pa_ctx_set_comm(ctx[r]->h, &comms[r]));
    }
    run_threads([&](int r) { ctx[r]->check(pa_comm_selftest(ctx[r]->h, 1 << 12)); });
  }

  template <class F>
  void run_threads(F f) {
    if (n == 1) { f(0); return; }
    std::vector<std::thread> th;
    for (int r = 0; r < n; ++r) th.emplace_back([&, r] { f(r); });
    for (auto& t : th) t.join();
  }
  // body(rank) on every rank; n = 1 runs on the calling thread
  template <class F>
  void run(F body) { run_threads(body); }

  // ---- in-process transport (pa_comm callbacks; errors abort the process like the tools' other failures)
  // PAIRWISE, like RCCL's grouped send / receive and the gloo transport: the library calls a rank's exchange only when
  // that rank has something to send or receive (pa_xexchange), so a rank that owns no box of a level -- or has no
  // neighbour on another rank -- never shows up and nothing here may wait for "all ranks".  Every ordered pair
  // (sender, receiver) has a FIFO mailbox; the k-th send of a to b meets the k-th receive of b from a (both sides derive
  // their lists from the same plans in the same order).  A rank posts all its sends first (never blocks), then serves
  // its receives (waits for the peer's post, copies device to device), then waits until its own posts were consumed,
  // so its packed buffers may be overwritten when the call returns.
  struct Post { const double* buf; long long n; bool done; };
  std::mutex xm;
  std::condition_variable xcv;
  std::vector<std::deque<Post*>> box;  // [sender * n + receiver]
  static int exchange(void* user, void*, int32_t cnt, const pa_xfer* x) {
    User* u = (User*)user;
    Team* T = u->t;
    const int me = u->r;
    T->ctx[me]->check(pa_sync(T->ctx[me]->h));  // my packed buffers are complete
    std::vector<Post> mine;
    mine.reserve(cnt);
    for (int i = 0; i < cnt; ++i)
      if (x[i].nsend > 0) mine.push_back(Post{x[i].sendbuf, x[i].nsend, false});
    {
      std::lock_guard<std::mutex> lk(T->xm);
      size_t k = 0;
      for (int i = 0; i < cnt; ++i)
        if (x[i].nsend > 0) {
          if (x[i].peer < 0 || x[i].peer >= T->n || x[i].peer == me) Abort("in-process transport: bad peer");
          T->box[(size_t)me * T->n + x[i].peer].push_back(&mine[k++]);
        }
    }
    T->xcv.notify_all();
    for (int i = 0; i < cnt; ++i) {
      if (x[i].nrecv <= 0) continue;
      const int p = x[i].peer;
      if (p < 0 || p >= T->n || p == me) Abort("in-process transport: bad peer");
      Post* q;
      {
        std::unique_lock<std::mutex> lk(T->xm);
        auto& mb = T->box[(size_t)p * T->n + me];
        T->xcv.wait(lk, [&] { return !mb.empty(); });
        q = mb.front();
        mb.pop_front();
      }
      if (q->n != x[i].nrecv) Abort("in-process transport: send / receive lists of two ranks do not match");
      T->ctx[me]->check(pa_memcpy_d2d(T->ctx[me]->h, x[i].recvbuf, q->buf, 8 * x[i].nrecv));  // synchronous
      {
        std::lock_guard<std::mutex> lk(T->xm);
        q->done = true;
      }
      T->xcv.notify_all();
    }
    {
      std::unique_lock<std::mutex> lk(T->xm);
      T->xcv.wait(lk, [&] { for (const Post& q : mine) if (!q.done) return false; return true; });
    }
    return 0;
  }
  static int allreduce(void* user, double* vals, int32_t cnt, int32_t op) {
    User* u = (User*)user;
    Team* T = u->t;
    T->red[u->r].assign(vals, vals + cnt);
    T->bar->wait();
    for (int i = 0; i < cnt; ++i) {
      double v = T->red[0][i];
      for (int r = 1; r < T->n; ++r) v = op == 0 ? std::min(v, T->red[r][i]) : (op == 1 ? std::max(v, T->red[r][i]) : v + T->red[r][i]);
      vals[i] = v;
    }
    T->bar->wait();
    return 0;
  }
};

// the team (HIP contexts, transport) brought up on a second thread while the caller reads the plotfile
struct AsyncTeam {
  std::future<std::unique_ptr<Team>> fut;
  std::unique_ptr<Team> team;
  explicit AsyncTeam(const ParmParse& pp) : fut(std::async(std::launch::async, [&pp] { return std::unique_ptr<Team>(new Team(pp)); })) {}
  Team& get() {
    if (!team) team = fut.get();
    return *team;
  }
};

// DistributionMapping(ba) (empty on one rank)
inline std::vector<int32_t> shard_boxes(const std::vector<Box3>& boxes, int nranks) {
  std::vector<int32_t> own;
  if (nranks <= 1) return own;
  std::vector<int32_t> b6(6 * boxes.size());
  for (size_t i = 0; i < boxes.size(); ++i)
    for (int d = 0; d < 3; ++d) { b6[6 * i + d] = boxes[i].lo[d]; b6[6 * i + 3 + d] = boxes[i].hi[d]; }
  own.resize(boxes.size());
  if (pa_distribution_map((int)own.size(), b6.data(), nranks, own.data()) != 0) Abort("pa_distribution_map failed");
  return own;
}
inline std::vector<std::vector<int32_t>> shard_levels(const PlotfileHeader& H, int nlev, int nranks) {
  std::vector<std::vector<int32_t>> own(nlev);
  for (int l = 0; l < nlev; ++l) own[l] = shard_boxes(H.lev[l].boxes, nranks);
  return own;
}

// a rank's boxes of a level (BoxArray order) and the copies between the whole level's host multifab and the rank's
struct Share {
  std::vector<int> gids;
  std::vector<Box3> boxes;
  Share(const std::vector<Box3>& all, const std::vector<int32_t>& owner, int rank) {
    for (size_t g = 0; g < all.size(); ++g)
      if (owner.empty() || owner[g] == rank) { gids.push_back((int)g); boxes.push_back(all[g]); }
  }
  // FAB chunks have the same size in both layouts (pa_mf_layout depends on the box, ncomp and ng only)
  void gather(const HostMF& glob, HostMF& loc) const {
    loc.define(boxes, glob.ncomp, glob.ng);
    for (size_t i = 0; i < gids.size(); ++i) std::memcpy(loc.data.data() + loc.off[i], glob.data.data() + glob.off[gids[i]], sizeof(double) * (size_t)(glob.ncomp * glob.cs[gids[i]]));
  }
  void scatter(const HostMF& loc, HostMF& glob) const {
    for (size_t i = 0; i < gids.size(); ++i) std::memcpy(glob.data.data() + glob.off[gids[i]], loc.data.data() + loc.off[i], sizeof(double) * (size_t)(glob.ncomp * glob.cs[gids[i]]));
  }
};

}  // namespace pa
