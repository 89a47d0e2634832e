// libmuscato_hip.so -- muscato's seed-and-extend hot path on MI355X (gfx950, wave64).
//
// Path (reference file:line, kshedden/muscato):
//   screen : cmd/muscato_screen/main.go:116-207 (read k-mer sketch), :256-366 (target scan)
//   join   : cmd/muscato/main.go:318-385 (sort) + cmd/muscato_confirm/main.go:375-416 (merge)
//   confirm: cmd/muscato_confirm/main.go:151-159 (cdiff), :171-250 (searchpairs)
//   select : cmd/muscato_combine_windows/main.go:36-60 (per-read best + MMTol)
//
// MI355X design: the target database stays resident in HBM as one 2-bit
// stream plus a k-mer -> (gene, offset) table of 64-byte buckets built once per (database,
// WindowWidth); reads are fixed-stride 2-bit records.  One pass = k_screen (window keys probe
// the table, candidates filtered from the index entry alone) -> k_confirm (XOR/popcount Hamming
// distance, then per-read best + MMTol and MaxMatches accounting in the same workgroup; bound by
// the cache lines it gathers) -> tile scan -> k_compact (tuples in read order).  Read prep
// (bytewise sort + collapse, muscato_prep.hpp) is a separate entry point.  The Bloom sketch of
// the reference only prunes work and cannot change results
// (SURVEY.md 8a note H): every candidate is verified exactly in k_confirm, including its
// window key.
//
// There is no CPU fallback in this library.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is resolved at run time (rccl_api)
#include <dlfcn.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/muscato_hip.h"

#include "kernels_common.hpp"
#include "kernels_index.hpp"
#include "kernels_screen.hpp"
#include "kernels_confirm.hpp"
#include "kernels_match.hpp"
#include "kernels_match_lane_inst.hpp"  // k_match_t: declaration only (defined in match_lane_rw*.hip)
#include "kernels_screen_lane.hpp"
MUSC_LANE_INSTANCES_4(extern)
MUSC_LANE_INSTANCES_8(extern)
MUSC_LANE_INSTANCES_12(extern)
MUSC_LANE_INSTANCES_8W(extern)
MUSC_LANE_INSTANCES_12W(extern)
MUSC_LANE_INSTANCES_16W(extern)
MUSC_LANE_INSTANCES_SPEC(extern)
MUSC_DMA_INSTANCES(extern)

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------

namespace {

thread_local std::string g_init_error;  // musc_init may run on one host thread per GPU

template <class T>
struct DevBuf {
  T* p = nullptr;
  uint64_t cap = 0;  // elements
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// roctx ranges around the kernel families (the reference's tracing hook is the
// CPUProfile flag, cmd/muscato_screen/main.go:530-538): visible in `rocprofv3 --marker-trace`.
// The marker library is resolved at run time; without it the ranges are no-ops.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push && pop) return;
        push = nullptr;
        pop = nullptr;
      }
    }
  }
};
const Roctx& roctx() {
  static const Roctx r;
  return r;
}
struct Range {
  bool on;
  explicit Range(const char* name) : on(roctx().push != nullptr) {
    if (on) (void)roctx().push(name);
  }
  ~Range() {
    if (on) (void)roctx().pop();
  }
};

// A stream capture in progress is ended on every exit path: an early return between
// hipStreamBeginCapture and hipStreamEndCapture would leave the stream capturing, and every later
// call on the context would fail until it is destroyed.
struct CaptureGuard {
  hipStream_t st = nullptr;
  bool active = false;
  hipError_t begin(hipStream_t s) {
    const hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
      st = s;
      active = true;
    }
    return e;
  }
  hipError_t end(hipGraph_t* g) {
    active = false;
    return hipStreamEndCapture(st, g);
  }
  ~CaptureGuard() {
    if (!active) return;
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(st, &g);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
  }
};

// temporary device allocations of one call, released on every exit path
struct TmpBufs {
  void* p[16] = {};
  int n = 0;
  template <class T>
  hipError_t alloc(T** out, size_t bytes) {
    void* q = nullptr;
    *out = nullptr;
    if (n >= 16) return hipErrorOutOfMemory;  // (more allocations than this helper was sized for)
    const hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
    if (e == hipSuccess) p[n++] = q;
    *out = (T*)q;
    return e;
  }
  void release() {
    for (int i = 0; i < n; i++) (void)hipFree(p[i]);
    n = 0;
  }
  ~TmpBufs() { release(); }
};

}  // namespace

// The MUSC_* environment knobs (tests, A/B runs, experiments), read ONCE per context at musc_init -- a pass
// never calls getenv -- and again only on musc_reload_env (tests that flip a knob on a live context).
struct EnvKnobs {
  enum { IDX_AUTO = 0, IDX_CLASSIC, IDX_LINES, IDX_CLASSIC64 };
  int index = IDX_AUTO;       // MUSC_INDEX = classic | lines | classic64: the two-kernel path (on that bucket layout)
  bool match_dma = false;     // MUSC_MATCH = dma: k_match_g (kernels_match_dma.hpp) where it is built for the run, k_match_t elsewhere
  bool screen_wg = false;     // MUSC_SCREEN = wg: k_screen on line buckets instead of k_screen_t
  int context = 0;            // MUSC_CONTEXT = narrow (1) | wide (2)
  bool no_x_context = false;  // MUSC_NO_X_CONTEXT
  bool force_wide = false;    // MUSC_DEBUG_FORCE_WIDE
  int index_bits = 0;         // MUSC_DEBUG_INDEX_BITS (0: not set)
  int debug_grid = 0;         // MUSC_DEBUG_GRID (0: not set)
  bool debug_sync = false;    // MUSC_DEBUG_SYNC
  int graph = -1;             // MUSC_GRAPH: -1 not set, else its value
  bool fused_compact = false; // MUSC_FUSED_COMPACT=1: the previous batch's tuples move into `hits` from inside the next batch's match launch (r02-r03's default); default: a k_compact_w per batch
  bool pipeline = false;      // MUSC_PIPELINE > 0
  bool no_spec = false;       // MUSC_NO_SPEC: never pick a geometry-specialised kernel instance
  int grid_rounds = 0;        // MUSC_GRID_ROUNDS: the fused kernels' grid = this many times the resident workgroups (0: the default)
  long batch_reads = 0;       // MUSC_BATCH_READS (0: not set)
  void read() {
    *this = EnvKnobs();
    auto is = [](const char* v, const char* w) { return v && !strcmp(v, w); };
    const char* e = getenv("MUSC_INDEX");
    index = is(e, "classic") ? IDX_CLASSIC : is(e, "lines") ? IDX_LINES : is(e, "classic64") ? IDX_CLASSIC64 : IDX_AUTO;
    e = getenv("MUSC_MATCH");
    match_dma = is(e, "dma");
    screen_wg = is(getenv("MUSC_SCREEN"), "wg");
    e = getenv("MUSC_CONTEXT");
    context = is(e, "narrow") ? 1 : is(e, "wide") ? 2 : 0;
    no_x_context = getenv("MUSC_NO_X_CONTEXT") != nullptr;
    force_wide = getenv("MUSC_DEBUG_FORCE_WIDE") != nullptr;
    if ((e = getenv("MUSC_DEBUG_INDEX_BITS"))) index_bits = atoi(e);
    if ((e = getenv("MUSC_DEBUG_GRID"))) debug_grid = atoi(e);
    debug_sync = getenv("MUSC_DEBUG_SYNC") != nullptr;
    if ((e = getenv("MUSC_GRAPH"))) graph = atoi(e);
    if ((e = getenv("MUSC_FUSED_COMPACT"))) fused_compact = atoi(e) > 0;
    if ((e = getenv("MUSC_PIPELINE"))) pipeline = atoi(e) > 0;
    no_spec = getenv("MUSC_NO_SPEC") != nullptr;
    if ((e = getenv("MUSC_GRID_ROUNDS"))) grid_rounds = atoi(e);
    if ((e = getenv("MUSC_BATCH_READS"))) batch_reads = atol(e);
  }
};

struct musc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  EnvKnobs env;

  // database
  uint32_t* db2 = nullptr;
  uint32_t* dbm2 = nullptr;  // null when the database holds no X (or an all-zero plane made for reads that do)
  bool db_has_x = false;     // the database holds an X
  bool reads_have_x = false; // some loaded read holds an X
  // reads with X on context buckets (k_match_t<.., XM = 1 | 2>): where each read's X are (k_read_xpos), and
  // whether the reads in hand fit that form under a given mismatch budget (k_xpos_check), cached
  DevBuf<uint32_t> rdx;
  uint64_t rdx_epoch = ~0ull;
  int rdx_wide = -1;  // the format of rdx: XPos<false> or XPos<true>
  uint64_t xok_epoch = ~0ull;
  double xok_pmatch = -1.0;
  int32_t xok_mmp1 = -1;
  bool xok = false;
  uint32_t* dbx = nullptr;   // with dbm2: one bit per 64-base block that holds an X
  uint64_t max_tlen = 0;     // longest target (context buckets flag an entry in bit 31 of its position)
  // a database with X on context buckets (k_match_t<.., XM = 2>): whether the reads in hand keep their X
  // out of the run's windows and within their xpos words (k_xpos_check_db), cached per read set and windows
  uint64_t xokdb_epoch = ~0ull;
  int32_t xokdb_key[CTX_MAX_W + 3] = {0};
  bool xokdb = false;
  uint64_t* seq_off = nullptr;
  uint32_t nseq = 0;
  uint64_t nbases = 0;
  uint64_t db_words = 0;

  // index
  int idx_ww = 0, idx_bits = 0, idx_direct = 0;
  int wide = 0;  // database >= 2^32 bases: 40-bit positions, gene numbers < 2^24
  Bucket* idx_T = nullptr;  // 2^idx_bits buckets: 64-byte Bucket, or (idx_lines) 128-byte LineBucket
  uint4* idx_E = nullptr;   // overflow entries
  bool idx_lines = false;   // the table holds line buckets (kernels_index.hpp)
  unsigned scrt_resident = 0;  // k_screen_t: waves resident at once (queried once per record stride)
  int scrt_rw = 0;
  uint64_t idx_T_bytes = 0, idx_E_cap = 0;  // allocated table bytes / entries (kept across rebuilds:
                                            // hipMalloc / hipFree of tens of GiB take seconds)
  uint64_t idx_n = 0;       // indexed window starts
  uint64_t idx_novf = 0;
  // index kind 1: context buckets (kernels_match.hpp); only one kind is resident at a time
  int idx_kind = 0;         // of the index idx_ww describes: 0 = Bucket table, 1 = CtxBucket table
  int idx_wide = 0;         // idx_kind 1: the buckets are CtxBucketW (200 bases of context, two inline entries)
  int idx_CL = 0;           // context buckets: bases of left context (the largest window start)
  CtxBucket* ctx_T = nullptr;
  CtxEntry* ctx_E = nullptr;
  uint64_t ctx_T_cap = 0, ctx_E_cap = 0;
  PathParams* d_pp = nullptr;   // k_screen / k_confirm / k_hot_probes: the run's parameters
  PathParams h_pp;              // what d_pp holds
  bool h_pp_valid = false;
  MatchParams* d_mp = nullptr;  // k_match's parameter block
  MatchParams h_mp;             // what d_mp holds
  bool h_mp_valid = false;
  int spec_geom = 0;            // the geometry-specialised k_match_t instance this pass launches (SpecGeom<n>), 0 = the general one
  DevBuf<uint4> spill;          // k_match: reported candidates beyond a tile's LDS list

  // reads
  uint32_t* rd = nullptr;
  uint32_t* rdm = nullptr;  // null when no read holds an X
  uint64_t nreads = 0;
  int rw = 0;
  uint32_t max_len = 0;

  // musc_reads_load_packed32(async): reads of one length on their way from the host -- the 2-bit
  // stream goes to `stage` in pieces on the copy stream s_up, an event per piece; a pass packs the
  // records of a batch (k_pack_reads_fixed) when the batch's pieces have arrived
  struct Upload {
    uint32_t* stage = nullptr;
    uint64_t stage_words = 0;
    hipStream_t s_up = nullptr;
    std::vector<hipEvent_t> ev;
    uint64_t piece = 0, n_pieces = 0;  // reads per piece, pieces of this upload
    uint64_t packed_upto = 0;          // reads whose records exist
    uint32_t L = 0;
    bool active = false;               // records are still missing
  } up;

  // per-batch work buffers
  // what k_screen hands to k_confirm, twice: in a pipelined pass k_screen fills one set on the
  // screen stream while k_confirm and k_compact drain the other on the confirm stream
  struct BatchSet {
    DevBuf<uint32_t> wb, rvalid, tbase, tcount;
    DevBuf<uint4> cdesc;
  } bs[2];
  int cur = 0;                     // the set the next launches use
  hipStream_t stream2 = nullptr;   // confirm stream of a pipelined pass
  hipStream_t s_confirm = nullptr; // where k_confirm .. k_advance go in the pass in flight
  hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr}, ev_join = nullptr;
  DevBuf<uint32_t> scan_tmp, tcount2, tpre;
  DevBuf<uint4> stage;
  // k_match_t moves the tuples a batch staged from inside the NEXT batch's launch: a second set
  DevBuf<uint32_t> tcount2_b, tpre_b;
  DevBuf<uint4> stage_b;
  DevBuf<uint32_t> p_nx;
  DevBuf<uint16_t> nmiss_tab;
  DevBuf<uint32_t> block_table;
  bool force_exact_blocks = false;
  // the MaxMatches screening of these reads, database and parameters was inconclusive once: later passes over the same
  // combination start with the exact per-block counters instead of screening, failing and repeating
  uint64_t exact_epoch = 0;
  musc_params exact_params;
  int cur_block_mode = 0;          // of the pass in flight
  uint32_t cur_block_thr = 0;
  PathParams last_pp;          // of the last musc_match_device
  uint32_t last_max_matches = 0;
  bool last_exact_blocks = false;  // block_table holds exact counters of that pass
  unsigned long long* counters = nullptr;  // [0] valid windows [1] accepted [2] hit cursor
  uint64_t* h_pinned = nullptr;            // 16 x u64 pinned staging

  DevBuf<musc_hit> hits;
  uint64_t nhits = 0;
  DevBuf<uint64_t> packed;      // staging of musc_hits_copy_packed / musc_hits_unpack for host pointers
  DevBuf<musc_hit> gathered;    // musc_gather_rccl: every context's tuples on this device
  uint32_t* d_flag = nullptr;   // one device word for kernels that report "does not fit"

  uint32_t batch_reads = 16u << 20;
  // A pass over the same reads, database and parameters as the last completed one needs no
  // sizing: its buffers are known to suffice, so it runs without host round trips.
  uint64_t data_epoch = 1;       // bumped whenever reads or database change
  uint64_t sized_epoch = 0;      // data_epoch of the last completed pass
  musc_params sized_params;      // its parameters
  bool sized_exact_blocks = false;
  uint32_t sized_bsz = 0;        // reads per batch it ended up with
  musc_stats stats;
  // MUSC_GRAPH=1: the sized pass on context buckets as a hipGraph (one launch instead of seven per
  // batch; no per-kernel timing in that mode)
  hipGraphExec_t graph_exec = nullptr;
  bool graph_failed = false;  // capture or instantiation failed once: sized passes run launch by launch
  uint64_t graph_epoch = 0;
  musc_params graph_params;
  int graph_block_mode = -1;
  uint32_t graph_batches = 0;
  // timing events are created once and reused by every pass (creating and destroying a pair per
  // kernel family per batch cost more than the kernels of a small pass)
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  // the mismatch-budget table on the device is valid for these inputs
  double nm_pmatch = -1.0;
  int32_t nm_mmp1 = -1;
  uint32_t nm_maxlen = 0xFFFFFFFFu;
};

namespace {

int fail(musc_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_init_error = buf;
  return code;
}

#define HIPCHK(c, expr)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail((c), 10, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

template <class T>
int ensure(musc_ctx* c, DevBuf<T>& b, uint64_t n, bool keep = false) {
  if (n <= b.cap && b.p) return 0;
  uint64_t ncap = std::max<uint64_t>(n, b.cap + b.cap / 2);
  ncap = std::max<uint64_t>(ncap, 1024);
  T* np = nullptr;
  HIPCHK(c, hipMalloc((void**)&np, ncap * sizeof(T) + 64));
  if (keep && b.p && b.cap) {
    hipError_t e = hipMemcpyAsync(np, b.p, b.cap * sizeof(T), hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
      (void)hipFree(np);
      return fail(c, 10, "hit buffer grow failed: %s", hipGetErrorString(e));
    }
  }
  b.release();
  b.p = np;
  b.cap = ncap;
  return 0;
}

// blocks of b threads covering n elements; a dispatch must stay below 2^32 work-items, callers
// with potentially larger n use grid-stride kernels instead
inline unsigned nblk(uint64_t n, unsigned b) { return (unsigned)((n + b - 1) / b); }

// u32 scan of n elements (in may equal out).  tmp must hold scan_tmp_elems(n).
uint64_t scan_tmp_elems(uint64_t n) {
  uint64_t t = 0;
  while (n > SCAN_TILE) {
    n = (n + SCAN_TILE - 1) / SCAN_TILE;
    t += (n + 8 + 3) & ~3ull;
  }
  return t + 8;
}

int scan_u32(musc_ctx* c, const uint32_t* in, uint32_t* out, uint64_t n, bool inclusive, uint32_t* tmp,
             hipStream_t st = nullptr) {
  if (!st) st = c->stream;
  const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb > 0x7FFFFFFFull) return fail(c, 11, "scan too large");
  if (nb <= 1) {
    if (inclusive) hipLaunchKernelGGL(k_scan_block<true>, dim3(1), dim3(SCAN_BLOCK), 0, st, in, out, (uint32_t*)nullptr, n);
    else hipLaunchKernelGGL(k_scan_block<false>, dim3(1), dim3(SCAN_BLOCK), 0, st, in, out, (uint32_t*)nullptr, n);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  uint32_t* sums = tmp;
  if (inclusive) hipLaunchKernelGGL(k_scan_block<true>, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, in, out, sums, n);
  else hipLaunchKernelGGL(k_scan_block<false>, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, in, out, sums, n);
  HIPCHK(c, hipGetLastError());
  int rc = scan_u32(c, sums, sums, nb, false, tmp + ((nb + 8 + 3) & ~3ull), st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, out, sums, n);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// exclusive u64 scan (in place allowed); tmp must hold scan_tmp_elems(n) u64
int scan_u64(musc_ctx* c, const uint64_t* in, uint64_t* out, uint64_t n, uint64_t* tmp) {
  const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb > 0x7FFFFFFFull) return fail(c, 11, "scan too large");
  if (nb <= 1) {
    hipLaunchKernelGGL(k_scan64_block, dim3(1), dim3(SCAN_BLOCK), 0, c->stream, in, out, (uint64_t*)nullptr, n);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(k_scan64_block, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, c->stream, in, out, tmp, n);
  HIPCHK(c, hipGetLastError());
  int rc = scan_u64(c, tmp, tmp, nb, tmp + ((nb + 8 + 3) & ~3ull));
  if (rc) return rc;
  hipLaunchKernelGGL(k_scan64_add, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, c->stream, out, tmp, n);
  HIPCHK(c, hipGetLastError());
  return 0;
}

void free_index(musc_ctx* c) {
  // (the allocations stay for the next build; musc_destroy releases them)
  c->idx_ww = 0;
  c->idx_n = 0;
  c->data_epoch++;
}

// release the context-bucket tables (the classic index is being built, or the database changes)
void drop_ctx_index(musc_ctx* c) {
  if (c->ctx_T) (void)hipFree(c->ctx_T);
  if (c->ctx_E) (void)hipFree(c->ctx_E);
  c->ctx_T = nullptr;
  c->ctx_E = nullptr;
  c->ctx_T_cap = c->ctx_E_cap = 0;
}

void free_db(musc_ctx* c) {
  free_index(c);
  if (c->db2) (void)hipFree(c->db2);
  if (c->dbm2) (void)hipFree(c->dbm2);
  if (c->dbx) (void)hipFree(c->dbx);
  if (c->seq_off) (void)hipFree(c->seq_off);
  c->db2 = c->dbm2 = c->dbx = nullptr;
  c->db_has_x = false;
  c->seq_off = nullptr;
  c->nseq = 0;
  c->nbases = 0;
  c->data_epoch++;
}

void free_reads(musc_ctx* c) {
  if (c->up.active) {  // an upload nobody matched: the caller's buffer is borrowed until it ends
    (void)hipStreamSynchronize(c->up.s_up);
    c->up.active = false;
  }
  if (c->rd) (void)hipFree(c->rd);
  if (c->rdm) (void)hipFree(c->rdm);
  c->rd = c->rdm = nullptr;
  c->reads_have_x = false;
  c->nreads = 0;
  c->rw = 0;
  c->data_epoch++;
}

struct EvPair {
  hipEvent_t a, b;
};

// an event of the context's pool (created on first use, destroyed with the context)
hipEvent_t pool_event(musc_ctx* c) {
  if (c->ev_used == c->ev_pool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    c->ev_pool.push_back(e);
  }
  return c->ev_pool[c->ev_used++];
}

// HIP-event timing of the kernel families of one pass; the events belong to the context's pool
struct Timer {
  musc_ctx* c;
  bool off = false;  // while a pass is captured into a hipGraph no events are recorded
  std::vector<EvPair> ev[5];
  explicit Timer(musc_ctx* ctx) : c(ctx) { c->ev_used = 0; }
  int begin(int fam, hipStream_t s = nullptr) {
    if (off) return 0;
    EvPair p;
    p.a = pool_event(c);
    p.b = pool_event(c);
    if (!p.a || !p.b) return 1;
    (void)hipEventRecord(p.a, s ? s : c->stream);
    ev[fam].push_back(p);
    return 0;
  }
  void end(int fam, hipStream_t s = nullptr) {
    if (!off && !ev[fam].empty()) (void)hipEventRecord(ev[fam].back().b, s ? s : c->stream);
  }
  float total(int fam) {
    float t = 0;
    for (auto& p : ev[fam]) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) t += ms;
    }
    return t;
  }
};

int check_params(musc_ctx* c, const musc_params* P) {
  if (!P) return fail(c, 2, "params is NULL");
  if (P->n_windows < 1 || P->n_windows > MUSC_MAX_WINDOWS)
    return fail(c, 2, "n_windows=%d outside 1..%d", P->n_windows, MUSC_MAX_WINDOWS);
  if (P->window_width < 1 || P->window_width > 4096) return fail(c, 2, "bad window_width=%d", P->window_width);
  for (int k = 0; k < P->n_windows; k++)
    if (P->windows[k] < 0 || P->windows[k] > 60000) return fail(c, 2, "bad window start %d", P->windows[k]);
  if (!(P->pmatch >= 0.0 && P->pmatch <= 1.0)) return fail(c, 2, "PMatch=%g outside [0,1]", P->pmatch);
  if (P->match_mode != 0 && P->match_mode != 1) return fail(c, 2, "match_mode must be 0 (best) or 1 (first)");
  if (P->mmtol < 0) return fail(c, 2, "MMTol < 0");
  if (P->max_mismatch_p1 < 0) return fail(c, 2, "max_mismatch_p1 < 0");
  return 0;
}

// line buckets without X anywhere, record strides k_screen_t is built for: the wave-autonomous screen
// (kernels_screen_lane.hpp); MUSC_SCREEN=wg keeps k_screen (A/B runs)
bool screen_lane(const musc_ctx* c, bool mask) {
  return c->idx_lines && !mask && !c->rdm && (c->rw == 4 || c->rw == 8 || c->rw == 12 || c->rw == 16) && !c->env.screen_wg;
}

// workgroups of the screen stage: the descriptor buffer is cut into that many regions.  k_screen_t's
// workgroups are single waves that stay for the whole launch: as many as are resident at once (a
// second, thinner round of them would cost what a full one does).
unsigned screen_grid(musc_ctx* c, uint32_t n, bool mask) {
  unsigned g = std::min(nblk(n, TILE), MAX_GRID);
  if (screen_lane(c, mask)) {
    if (!c->scrt_resident) {
      int per_cu = 0, ncu = 0;
      const void* fn = c->rw == 4 ? reinterpret_cast<const void*>(&k_screen_t<4>) : c->rw == 8 ? reinterpret_cast<const void*>(&k_screen_t<8>)
                       : c->rw == 12 ? reinterpret_cast<const void*>(&k_screen_t<12>) : reinterpret_cast<const void*>(&k_screen_t<16>);
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 8; }
      if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || ncu < 1) ncu = 256;
      c->scrt_resident = (unsigned)per_cu * (unsigned)ncu;
      c->scrt_rw = c->rw;
    }
    if (c->scrt_rw != c->rw) { c->scrt_resident = 0; return screen_grid(c, n, mask); }
    g = std::min(g, std::min(c->scrt_resident, MAX_GRID));
  }
  return g;
}

template <int RW>
void launch_path(musc_ctx* c, int stage, bool mask, uint64_t r0, uint32_t n, const PathParams& pp) {
  const dim3 block(TILE);
  if (stage == 0) {
    const dim3 sgrid(screen_grid(c, n, mask));
    if constexpr (RW == 4 || RW == 8 || RW == 12 || RW == 16) {
      if (screen_lane(c, mask)) {
        hipLaunchKernelGGL((k_screen_t<RW>), sgrid, dim3(64), 0, c->stream, c->rd, r0, n, c->d_pp, c->nmiss_tab.p,
                           reinterpret_cast<const LineBucket*>(c->idx_T), c->idx_E, c->bs[c->cur].cdesc.p, c->bs[c->cur].cdesc.cap,
                           c->bs[c->cur].rvalid.p, c->bs[c->cur].wb.p, c->bs[c->cur].tbase.p, c->bs[c->cur].tcount.p,
                           c->counters + 8, c->counters + 3);
        return;
      }
    }
#define MUSC_LAUNCH_SCREEN(M, O)                                                                                  \
    do { if (c->idx_lines) MUSC_LAUNCH_SCREEN2(M, O, true); else MUSC_LAUNCH_SCREEN2(M, O, false); } while (0)
#define MUSC_LAUNCH_SCREEN2(M, O, LN)                                                                             \
    hipLaunchKernelGGL((k_screen<RW, M, O, LN>), sgrid, block, 0, c->stream, c->rd, c->rdm, r0, n, c->rw, c->d_pp, \
                       c->nmiss_tab.p, c->idx_T, c->idx_E, c->bs[c->cur].cdesc.p, c->bs[c->cur].cdesc.cap,        \
                       c->bs[c->cur].rvalid.p, c->bs[c->cur].wb.p, c->bs[c->cur].tbase.p, c->bs[c->cur].tcount.p, \
                       c->counters + 8, c->counters + 3)
    const bool one = pp.W <= 2;
    if (c->rdm) { if (one) MUSC_LAUNCH_SCREEN(true, true); else MUSC_LAUNCH_SCREEN(true, false); }
    else { if (one) MUSC_LAUNCH_SCREEN(false, true); else MUSC_LAUNCH_SCREEN(false, false); }
#undef MUSC_LAUNCH_SCREEN
#undef MUSC_LAUNCH_SCREEN2
  } else {
    // persistent over tiles; the MaxMatches screening threshold assumes at most MAX_GRID workgroups
    const dim3 grid(std::min(nblk(n, TILE), MAX_GRID));
    static_assert((1u << 24) / TILE / MAX_GRID <= CONF_TILES, "a k_confirm workgroup keeps its tile list in LDS");
    const size_t lds = c->cur_block_mode ? (size_t)TILE * pp.W * 4 : 0;
#define MUSC_LAUNCH_CONFIRM(M, W2)                                                                                 \
    hipLaunchKernelGGL((k_confirm<RW, M, W2>), grid, block, lds, c->s_confirm, c->rd, c->rdm, c->db2, c->dbm2,      \
                       c->dbx, r0, n, c->rw, c->d_pp, c->nmiss_tab.p, c->bs[c->cur].cdesc.p, c->bs[c->cur].rvalid.p,     \
                       c->p_nx.p, c->bs[c->cur].tbase.p, c->bs[c->cur].tcount.p, c->bs[c->cur].wb.p,                \
                       c->cur_block_mode, c->cur_block_thr, c->block_table.p, c->seq_off, c->stage.p, c->tcount2.p, \
                       c->counters)
    const bool w2 = pp.W <= 2;
    if (mask) { if (w2) MUSC_LAUNCH_CONFIRM(true, true); else MUSC_LAUNCH_CONFIRM(true, false); }
    else { if (w2) MUSC_LAUNCH_CONFIRM(false, true); else MUSC_LAUNCH_CONFIRM(false, false); }
#undef MUSC_LAUNCH_CONFIRM
  }
}

void launch_stage(musc_ctx* c, int stage, bool mask, uint64_t r0, uint32_t n, const PathParams& pp) {
  switch (c->rw) {
    case 4: launch_path<4>(c, stage, mask, r0, n, pp); break;
    case 8: launch_path<8>(c, stage, mask, r0, n, pp); break;
    case 12: launch_path<12>(c, stage, mask, r0, n, pp); break;
    case 16: launch_path<16>(c, stage, mask, r0, n, pp); break;
    default: launch_path<0>(c, stage, mask, r0, n, pp); break;
  }
}

}  // namespace

extern "C" {

int musc_abi_version(void) { return MUSC_ABI_VERSION; }

const char* musc_last_error(musc_ctx* ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

int musc_init(int device_ordinal, musc_ctx** out) {
  if (!out) return fail(nullptr, 2, "musc_init: out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, 3, "musc_init: no HIP device (%s); this library has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device_ordinal < 0 || device_ordinal >= ndev)
    return fail(nullptr, 2, "musc_init: device %d outside 0..%d", device_ordinal, ndev - 1);
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device_ordinal)) != hipSuccess)
    return fail(nullptr, 3, "musc_init: hipGetDeviceProperties: %s", hipGetErrorString(e));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, 3, "musc_init: device %d is %s; this library is built for gfx950 only",
                device_ordinal, prop.gcnArchName);
  if ((e = hipSetDevice(device_ordinal)) != hipSuccess)
    return fail(nullptr, 3, "musc_init: hipSetDevice: %s", hipGetErrorString(e));
  musc_ctx* c = new musc_ctx();
  c->device = device_ordinal;
  memset(&c->stats, 0, sizeof c->stats);
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_ready[0], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_ready[1], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_free[0], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_free[1], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)) != hipSuccess ||
      (e = hipMalloc((void**)&c->counters, 16 * sizeof(unsigned long long))) != hipSuccess ||
      (e = hipMalloc((void**)&c->d_flag, 4)) != hipSuccess ||
      (e = hipMalloc((void**)&c->d_mp, sizeof(MatchParams))) != hipSuccess ||
      (e = hipMalloc((void**)&c->d_pp, sizeof(PathParams))) != hipSuccess ||
      (e = hipHostMalloc((void**)&c->h_pinned, 16 * sizeof(uint64_t))) != hipSuccess) {
    fail(nullptr, 3, "musc_init: %s", hipGetErrorString(e));
    musc_destroy(c);
    return 3;
  }
  c->env.read();
  if (c->env.batch_reads >= 1 && c->env.batch_reads <= (16l << 20)) c->batch_reads = (uint32_t)c->env.batch_reads;  // tests: many small batches
  *out = c;
  return 0;
}

// Re-read the MUSC_* knobs (musc_init reads them once; MUSC_BATCH_READS stays as it was read then).  A test hook:
// a knob that changes which index or kernel a pass takes makes the next pass size itself again.
int musc_reload_env(musc_ctx* c) {
  if (!c) return 1;
  c->env.read();
  c->sized_epoch = 0;
  return 0;
}

void musc_destroy(musc_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  free_db(c);
  free_reads(c);
  if (c->idx_T) (void)hipFree(c->idx_T);
  if (c->idx_E) (void)hipFree(c->idx_E);
  if (c->ctx_T) (void)hipFree(c->ctx_T);
  if (c->ctx_E) (void)hipFree(c->ctx_E);
  if (c->d_mp) (void)hipFree(c->d_mp);
  if (c->d_pp) (void)hipFree(c->d_pp);
  c->spill.release();
  for (int i = 0; i < 2; i++) {
    c->bs[i].wb.release(); c->bs[i].tbase.release(); c->bs[i].rvalid.release(); c->bs[i].tcount.release();
    c->bs[i].cdesc.release();
  }
  c->scan_tmp.release(); c->tcount2.release(); c->tpre.release(); c->stage.release(); c->rdx.release();
  c->tcount2_b.release(); c->tpre_b.release(); c->stage_b.release();
  c->p_nx.release();
  c->nmiss_tab.release();
  c->block_table.release();
  c->hits.release();
  c->packed.release();
  c->gathered.release();
  if (c->d_flag) (void)hipFree(c->d_flag);
  if (c->counters) (void)hipFree(c->counters);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  for (hipEvent_t ev : {c->ev_ready[0], c->ev_ready[1], c->ev_free[0], c->ev_free[1], c->ev_join})
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : c->ev_pool) (void)hipEventDestroy(ev);
  if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
  for (hipEvent_t ev : c->up.ev) (void)hipEventDestroy(ev);
  if (c->up.stage) (void)hipFree(c->up.stage);
  if (c->up.s_up) (void)hipStreamDestroy(c->up.s_up);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// ---------------------------------------------------------------- database

// the X-block bitmap of the mask plane (all zero for a plane that exists only because reads hold X)
static int db_xblocks(musc_ctx* c) {
  const uint64_t nblk64 = (c->db_words + 64 + 3) / 4;
  const uint64_t nw = nblk64 / 32 + 4;
  if (c->dbx) (void)hipFree(c->dbx);
  c->dbx = nullptr;
  HIPCHK(c, hipMalloc((void**)&c->dbx, nw * 4));
  HIPCHK(c, hipMemsetAsync(c->dbx, 0, nw * 4, c->stream));
  hipLaunchKernelGGL(k_db_xblocks, dim3(nblk(nblk64, 256)), dim3(256), 0, c->stream, c->dbm2, c->db_words, c->dbx);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

static int db_finish(musc_ctx* c) {
  uint32_t hasx = 0;
  HIPCHK(c, hipMemcpyAsync(&hasx, c->d_flag, 4, hipMemcpyDeviceToHost, c->stream));
  // the longest target
  unsigned long long tl = 0;
  HIPCHK(c, hipMemsetAsync(c->counters + 4, 0, 8, c->stream));
  hipLaunchKernelGGL(k_max_len, dim3(std::min(nblk(c->nseq, 256), MAX_GRID)), dim3(256), 0, c->stream, c->seq_off, (uint64_t)c->nseq, c->counters + 4);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(&tl, c->counters + 4, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->max_tlen = tl;
  c->db_has_x = hasx != 0;
  if (!hasx) {
    (void)hipFree(c->dbm2);
    c->dbm2 = nullptr;
    return 0;
  }
  return db_xblocks(c);
}

static int db_alloc(musc_ctx* c, const uint64_t* offsets, uint32_t nseq, int on_device) {
  HIPCHK(c, hipSetDevice(c->device));
  free_db(c);
  if (nseq == 0) return fail(c, 2, "database has no sequences");
  uint64_t first = 0, last = 0;
  if (on_device) {
    HIPCHK(c, hipMemcpy(&first, offsets, 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&last, offsets + nseq, 8, hipMemcpyDeviceToHost));
  } else {
    first = offsets[0];
    last = offsets[nseq];
  }
  if (first != 0) return fail(c, 2, "offsets[0] must be 0");
  if (last >= (1ull << 36)) return fail(c, 2, "database larger than 2^36 bases");
  c->nseq = nseq;
  c->nbases = last;
  c->db_words = (last + 15) / 16;
  const uint64_t alloc_words = c->db_words + 64;  // slack: k_confirm reads a whole span past the last base
  HIPCHK(c, hipMalloc((void**)&c->db2, alloc_words * 4));
  HIPCHK(c, hipMalloc((void**)&c->dbm2, alloc_words * 4));
  HIPCHK(c, hipMemsetAsync(c->db2, 0, alloc_words * 4, c->stream));
  HIPCHK(c, hipMemsetAsync(c->dbm2, 0, alloc_words * 4, c->stream));
  HIPCHK(c, hipMalloc((void**)&c->seq_off, ((uint64_t)nseq + 1) * 8));
  HIPCHK(c, hipMemcpyAsync(c->seq_off, offsets, ((uint64_t)nseq + 1) * 8,
                           on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_flag, 0, 4, c->stream));  // raised by the packing kernel when it meets an X
  return 0;
}

int musc_db_load_ascii(musc_ctx* c, const char* seqs, const uint64_t* offsets, uint32_t nseq, int on_device) {
  if (!c) return 1;
  if (!seqs || !offsets) return fail(c, 2, "musc_db_load_ascii: NULL input");
  int rc = db_alloc(c, offsets, nseq, on_device);
  if (rc) return rc;
  const unsigned char* src = (const unsigned char*)seqs;
  TmpBufs B;
  unsigned char* tmp = nullptr;
  if (!on_device && c->nbases) {
    HIPCHK(c, B.alloc(&tmp, c->nbases));
    HIPCHK(c, hipMemcpyAsync(tmp, seqs, c->nbases, hipMemcpyHostToDevice, c->stream));
    src = tmp;
  }
  if (c->db_words) {
    hipLaunchKernelGGL(k_pack_db_ascii, dim3(nblk(c->db_words, 256)), dim3(256), 0, c->stream, src, c->nbases,
                       c->db2, c->dbm2, c->db_words, c->d_flag);
    HIPCHK(c, hipGetLastError());
  }
  return db_finish(c);
}

int musc_db_load_packed(musc_ctx* c, const uint8_t* bases2bit, const uint8_t* nmask, const uint64_t* seq_offsets,
                        uint32_t nseq) {
  if (!c) return 1;
  if (!bases2bit || !seq_offsets) return fail(c, 2, "musc_db_load_packed: NULL input");
  int rc = db_alloc(c, seq_offsets, nseq, 0);
  if (rc) return rc;
  TmpBufs B;
  uint32_t* t2 = nullptr;
  uint16_t* tm = nullptr;
  const uint64_t w = c->db_words;
  if (w) {
    HIPCHK(c, B.alloc(&t2, w * 4));
    HIPCHK(c, hipMemsetAsync(t2, 0, w * 4, c->stream));
    HIPCHK(c, hipMemcpyAsync(t2, bases2bit, (c->nbases + 3) / 4, hipMemcpyHostToDevice, c->stream));
    if (nmask) {
      HIPCHK(c, B.alloc(&tm, w * 2));
      HIPCHK(c, hipMemsetAsync(tm, 0, w * 2, c->stream));
      HIPCHK(c, hipMemcpyAsync(tm, nmask, (c->nbases + 7) / 8, hipMemcpyHostToDevice, c->stream));
    }
    hipLaunchKernelGGL(k_pack_db_packed, dim3(nblk(w, 256)), dim3(256), 0, c->stream, t2, tm, c->db2, c->dbm2, w,
                       c->d_flag);
    HIPCHK(c, hipGetLastError());
  }
  return db_finish(c);
}

extern "C++" {
// Which bucket layout the two-kernel index uses for this database and window width: line buckets
// (LineBucket: a 128-byte line of seven entries + aligned overflow runs) when the direct table has
// four or more window starts per key on average and the table fits, 64-byte buckets otherwise.
// MUSC_INDEX=lines / classic64 force one or the other.
static bool want_line_buckets(musc_ctx* c, int32_t ww, int bits, int direct) {
  if (c->env.index == EnvKnobs::IDX_LINES) return true;
  if (c->env.index == EnvKnobs::IDX_CLASSIC64) return false;
  if (!direct) return false;
  (void)ww;
  const uint64_t nb = 1ull << bits;
  if (c->nbases < 4 * nb) return false;
  size_t mfree = 0, mtotal = 0;
  if (hipMemGetInfo(&mfree, &mtotal) != hipSuccess) return false;
  const uint64_t have = (uint64_t)mfree + c->idx_T_bytes + c->idx_E_cap * sizeof(uint4) +
                        (c->ctx_T ? c->ctx_T_cap * sizeof(CtxBucket) : 0) + (c->ctx_E ? c->ctx_E_cap * sizeof(CtxEntry) : 0);
  // the table, the entries beyond the seventh in runs of eight (assume every bucket wastes half a run),
  // 8 B per bucket of build temporaries, and room for the pass's buffers
  const uint64_t need = (nb + 1) * (sizeof(LineBucket) + 8) + (c->nbases > 7 * nb ? (c->nbases - 7 * nb) * 16 : 0) + nb * 64 + (12ull << 30);
  return need <= have;
}

template <class BT>
static int build_index_buckets(musc_ctx* c, int32_t ww, int bits, int direct) {
  const uint64_t nb = 1ull << bits;
  c->ev_used = 0;
  hipEvent_t e0 = pool_event(c), e1 = pool_event(c), e2 = pool_event(c), e3 = pool_event(c);
  if (!e0 || !e1 || !e2 || !e3) return fail(c, 10, "hipEventCreate failed");
  if (c->idx_T_bytes < (nb + 1) * sizeof(BT)) {
    if (c->idx_T) (void)hipFree(c->idx_T);
    c->idx_T = nullptr;
    c->idx_T_bytes = 0;
    HIPCHK(c, hipMalloc((void**)&c->idx_T, (nb + 1) * sizeof(BT)));
    c->idx_T_bytes = (nb + 1) * sizeof(BT);
  }
  BT* const T = reinterpret_cast<BT*>(c->idx_T);
  TmpBufs B;
  uint64_t *tmp = nullptr, *stmp = nullptr;
  HIPCHK(c, B.alloc(&tmp, (nb + 1 + 16) * 8));
  HIPCHK(c, B.alloc(&stmp, scan_tmp_elems(nb + 1) * 8));
  // timed: the device work (allocation above and below is host time, seconds for a 64 GiB table
  // the first time, and not repeated)
  HIPCHK(c, hipEventRecord(e0, c->stream));
  HIPCHK(c, hipMemsetAsync(T, 0, (nb + 1) * sizeof(BT), c->stream));
  const unsigned blocks = (unsigned)std::min<uint64_t>((c->nbases + 255) / 256, 1u << 22);
  if (c->nbases) {
    hipLaunchKernelGGL((k_index<false, BT>), dim3(blocks), dim3(256), 0, c->stream, c->db2, c->dbm2, c->seq_off, c->nseq,
                       c->nbases, ww, bits, direct, c->wide, T, (uint4*)nullptr);
    HIPCHK(c, hipGetLastError());
  }
  // overflow lists: sizes -> offsets (u64: a 10 Gbp database has billions of overflow entries)
  hipLaunchKernelGGL((k_index_ovf_count<BT>), dim3(nblk(nb + 1, 256)), dim3(256), 0, c->stream, T, nb, tmp);
  HIPCHK(c, hipGetLastError());
  int rc = scan_u64(c, tmp, tmp, nb + 1, stmp);
  if (rc) return rc;
  uint64_t novf = 0;  // entries (64-byte buckets) or lines of eight entries (line buckets)
  HIPCHK(c, hipMemcpyAsync(&novf, tmp + nb, 8, hipMemcpyDeviceToHost, c->stream));
  hipLaunchKernelGGL((k_index_ovf_set<BT>), dim3(nblk(nb, 256)), dim3(256), 0, c->stream, T, nb, tmp);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(e2, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  B.release();  // 8 bytes per bucket: returned before the overflow array is allocated
  if (std::is_same<BT, LineBucket>::value) {
    if (novf >= 0xFFFFFFF0ull) return fail(c, 5, "internal: %llu overflow lines do not fit 32-bit numbers", (unsigned long long)novf);
    novf *= 8;
  }
  c->idx_novf = novf;
  if (c->idx_E_cap < novf + 16) {
    if (c->idx_E) (void)hipFree(c->idx_E);
    c->idx_E = nullptr;
    c->idx_E_cap = 0;
    HIPCHK(c, hipMalloc((void**)&c->idx_E, ((uint64_t)novf + 16) * sizeof(uint4)));
    c->idx_E_cap = novf + 16;
  }
  HIPCHK(c, hipEventRecord(e3, c->stream));
  if (c->nbases) {
    hipLaunchKernelGGL((k_index<true, BT>), dim3(blocks), dim3(256), 0, c->stream, c->db2, c->dbm2, c->seq_off, c->nseq,
                       c->nbases, ww, bits, direct, c->wide, T, c->idx_E);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float ms = 0, ms2 = 0;
  (void)hipEventElapsedTime(&ms, e0, e2);
  (void)hipEventElapsedTime(&ms2, e3, e1);
  c->stats.ms_index_build = ms + ms2;
  return 0;
}
}  // extern "C++"

int musc_db_build_index(musc_ctx* c, int32_t ww) {
  if (!c) return 1;
  if (!c->db2) return fail(c, 4, "no database loaded");
  if (ww < 1 || ww > 4096) return fail(c, 2, "bad window width %d", ww);
  HIPCHK(c, hipSetDevice(c->device));
  c->wide = c->nbases >= 0xFFFFFFF0ull || c->env.force_wide;
  if (c->wide && c->nseq >= (1u << 24))
    return fail(c, 5, "a database of 2^32 bases or more may hold at most 2^24 targets (has %u)", c->nseq);
  // Direct addressing (bucket = the 2*ww-bit key itself: exact, and bytewise-sorted reads walk
  // the table front to back) when that table is at most 32x the database and at most 2^30
  // buckets (64 GiB; line buckets: 128 GiB); otherwise a hashed table with about one bucket per base, at
  // most 2^31 buckets (128 GiB; longer lists go to the overflow array).
  int bits, direct = 0;
  const uint64_t floor_bases = std::max<uint64_t>(c->nbases, 1ull << 19);
  if (2 * ww <= 30 && (1ull << (2 * ww)) <= 32 * floor_bases) {
    bits = 2 * ww;
    direct = 1;
  } else {
    bits = 10;
    while (bits < 31 && (1ull << bits) < c->nbases) bits++;
  }
  if (c->env.index_bits >= 8 && c->env.index_bits <= 31) { bits = c->env.index_bits; direct = 0; }  // experiments only: force a hashed table size
  // A resident window-start index for this width keeps its layout: the automatic choice looks at the free memory
  // of the moment (want_line_buckets), which moves as the pass buffers grow, and a flip would mean dropping the
  // sized state and rebuilding tens of gigabytes in the middle of a run.  Only an explicit MUSC_INDEX = lines |
  // classic64 that contradicts the resident layout rebuilds.
  if (c->idx_ww == ww && c->idx_kind == 0 && c->idx_T && c->idx_bits == bits && c->idx_direct == direct &&
      !(c->env.index == EnvKnobs::IDX_LINES && !c->idx_lines) && !(c->env.index == EnvKnobs::IDX_CLASSIC64 && c->idx_lines))
    return 0;
  const bool lines = want_line_buckets(c, ww, bits, direct) && bits <= 30;
  free_index(c);
  drop_ctx_index(c);  // one index kind is resident at a time
  c->idx_kind = 0;
  c->idx_lines = lines;
  const int rc = lines ? build_index_buckets<LineBucket>(c, ww, bits, direct) : build_index_buckets<Bucket>(c, ww, bits, direct);
  if (rc) return rc;
  c->idx_ww = ww;
  c->idx_bits = bits;
  c->idx_direct = direct;
  return 0;
}

// Context buckets for window width ww and CL bases of left context (kernels_match.hpp).
// Returns 0 and leaves idx_kind == 1 on success; 100 when the table does not fit the device's
// free memory (the caller then builds the classic index); anything else is an error.
static int build_index_ctx(musc_ctx* c, int32_t ww, int32_t CL, int wide) {
  if (c->idx_kind == 1 && c->idx_ww == ww && c->idx_CL == CL && c->idx_wide == wide && c->ctx_T) return 0;
  // (ctx_E_cap counts 40-byte CtxEntry; a wide index asks for its 60-byte entries in those units)
  const uint64_t esz = wide ? sizeof(CtxEntryW) : sizeof(CtxEntry);
  free_index(c);
  c->wide = 0;
  // 4^ww buckets with the key as the bucket (exact) when that is at most twice the database's
  // window count; else about one bucket per base under a 64-bit mix (a colliding key fails the
  // window comparison in k_match: the context includes the window bases)
  int bits, direct = 0;
  const uint64_t floor_bases = std::max<uint64_t>(c->nbases, 1ull << 9);
  if (2 * ww <= 30 && (1ull << (2 * ww)) <= 2 * floor_bases) {
    bits = 2 * ww;
    direct = 1;
  } else {
    bits = 10;
    while (bits < 30 && (1ull << bits) < c->nbases) bits++;
  }
  if (c->env.index_bits >= 8 && c->env.index_bits <= 30) { bits = c->env.index_bits; direct = 0; }  // experiments only: force a hashed table size
  const uint64_t nb = 1ull << bits;
  // memory: the table, 8 B + 4 B per bucket of build temporaries, and the overflow entries (their
  // number is known only after the counting pass: assume a third of the windows for the estimate)
  {
    size_t mfree = 0, mtotal = 0;
    HIPCHK(c, hipMemGetInfo(&mfree, &mtotal));
    const uint64_t have = (uint64_t)mfree + (c->ctx_T ? c->ctx_T_cap * sizeof(CtxBucket) : 0) +
                          (c->ctx_E ? c->ctx_E_cap * sizeof(CtxEntry) : 0) + c->idx_T_bytes +
                          (c->idx_E ? c->idx_E_cap * sizeof(uint4) : 0);
    const uint64_t need = (nb + 1) * (sizeof(CtxBucket) + 12) + c->nbases / (wide ? 2 : 3) * esz + (4ull << 30);
    if (need > have) return 100;
  }
  // the classic tables go first (one index kind is resident at a time)
  if (c->idx_T) (void)hipFree(c->idx_T);
  if (c->idx_E) (void)hipFree(c->idx_E);
  c->idx_T = nullptr;
  c->idx_E = nullptr;
  c->idx_T_bytes = c->idx_E_cap = 0;
  c->ev_used = 0;
  hipEvent_t e0 = pool_event(c), e1 = pool_event(c), e2 = pool_event(c), e3 = pool_event(c);
  if (!e0 || !e1 || !e2 || !e3) return fail(c, 10, "hipEventCreate failed");
  if (c->ctx_T_cap < nb + 1) {
    if (c->ctx_T) (void)hipFree(c->ctx_T);
    c->ctx_T = nullptr;
    c->ctx_T_cap = 0;
    if (hipMalloc((void**)&c->ctx_T, (nb + 1) * sizeof(CtxBucket)) != hipSuccess) {
      (void)hipGetLastError();
      c->ctx_T = nullptr;
      return 100;
    }
    c->ctx_T_cap = nb + 1;
  }
  TmpBufs B;
  uint64_t *tmp = nullptr, *stmp = nullptr;
  uint32_t* cursor = nullptr;
  if (B.alloc(&tmp, (nb + 1 + 16) * 8) != hipSuccess || B.alloc(&stmp, scan_tmp_elems(nb + 1) * 8) != hipSuccess ||
      B.alloc(&cursor, (nb + 1) * 4) != hipSuccess) {
    (void)hipGetLastError();
    return 100;
  }
  HIPCHK(c, hipEventRecord(e0, c->stream));
  HIPCHK(c, hipMemsetAsync(c->ctx_T, 0, (nb + 1) * sizeof(CtxBucket), c->stream));
  HIPCHK(c, hipMemsetAsync(cursor, 0, (nb + 1) * 4, c->stream));
  const unsigned blocks = (unsigned)std::min<uint64_t>((c->nbases + 255) / 256, 1u << 22);
  // a database with X: its windows with an X stay out, entries whose context touches one are flagged
  const uint32_t* const xm2 = c->db_has_x ? c->dbm2 : nullptr;
  const uint32_t* const xbl = c->db_has_x ? c->dbx : nullptr;
  if (c->nbases) {
    // (the counting pass does not look at the entries: one instance serves both layouts)
    hipLaunchKernelGGL((k_index_ctx<false, false>), dim3(blocks), dim3(256), 0, c->stream, c->db2, xm2, xbl, c->seq_off, c->nseq, c->nbases,
                       ww, bits, direct, CL, c->ctx_T, (void*)nullptr, cursor);
    HIPCHK(c, hipGetLastError());
  }
  hipLaunchKernelGGL(k_ctx_ovf_count, dim3(nblk(nb + 1, 256)), dim3(256), 0, c->stream, c->ctx_T, nb,
                     (uint32_t)(wide ? CTXW_INLINE : CTX_INLINE), tmp);
  HIPCHK(c, hipGetLastError());
  int rc = scan_u64(c, tmp, tmp, nb + 1, stmp);
  if (rc) return rc;
  uint64_t novf = 0;
  HIPCHK(c, hipMemcpyAsync(&novf, tmp + nb, 8, hipMemcpyDeviceToHost, c->stream));
  hipLaunchKernelGGL(k_ctx_ovf_set, dim3(nblk(nb, 256)), dim3(256), 0, c->stream, c->ctx_T, nb, tmp);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(e2, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (novf >= 0xFFFFFFF0ull) return fail(c, 5, "internal: %llu overflow entries do not fit 32-bit offsets", (unsigned long long)novf);
  c->idx_novf = novf;
  const uint64_t ecap = (ctx_entries_bytes(novf + 16, wide) + sizeof(CtxEntry) - 1) / sizeof(CtxEntry);  // (entries sit line-aligned: ctx_entry_word)
  if (c->ctx_E_cap < ecap) {
    if (c->ctx_E) (void)hipFree(c->ctx_E);
    c->ctx_E = nullptr;
    c->ctx_E_cap = 0;
    if (hipMalloc((void**)&c->ctx_E, ecap * sizeof(CtxEntry)) != hipSuccess) {
      (void)hipGetLastError();
      c->ctx_E = nullptr;
      return 100;
    }
    c->ctx_E_cap = ecap;
  }
  HIPCHK(c, hipEventRecord(e3, c->stream));
  if (c->nbases) {
    if (wide)
      hipLaunchKernelGGL((k_index_ctx<true, true>), dim3(blocks), dim3(256), 0, c->stream, c->db2, xm2, xbl, c->seq_off, c->nseq, c->nbases,
                         ww, bits, direct, CL, c->ctx_T, (void*)c->ctx_E, cursor);
    else
      hipLaunchKernelGGL((k_index_ctx<true, false>), dim3(blocks), dim3(256), 0, c->stream, c->db2, xm2, xbl, c->seq_off, c->nseq, c->nbases,
                         ww, bits, direct, CL, c->ctx_T, (void*)c->ctx_E, cursor);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float ms = 0, ms2 = 0;
  (void)hipEventElapsedTime(&ms, e0, e2);
  (void)hipEventElapsedTime(&ms2, e3, e1);
  c->stats.ms_index_build = ms + ms2;
  c->idx_kind = 1;
  c->idx_wide = wide;
  c->idx_ww = ww;
  c->idx_CL = CL;
  c->idx_bits = bits;
  c->idx_direct = direct;
  return 0;
}

// Which of the two fused kernels on context buckets runs
#ifndef MATCH_GRID_ROUNDS
#define MATCH_GRID_ROUNDS 1
#endif
enum MatchKind { MK_LANE = 2, MK_DMA = 3 };
// MK_LANE = k_match_t (kernels_match_lane.hpp): every run on context buckets.  MK_DMA = k_match_g
// (kernels_match_dma.hpp): the same comparisons at three to four waves per SIMD, everything from memory by LDS-DMA --
// built for two windows on 120-base buckets, records of 8 words, no X on either side (BASELINE configs 2-4).  It is
// the second implementation (MUSC_MATCH=dma; the parity tests run both): on cfg3 its launch takes as long as
// k_match_t's.
static int match_kind(const musc_ctx* c, int W) {
  if (c->env.match_dma && W == 2 && c->rw == 8 && c->idx_kind == 1 && !c->idx_wide && !c->db_has_x && !c->reads_have_x) return MK_DMA;
  return MK_LANE;
}

// The xpos words of the reads in hand, in the format of the bucket width
static bool reads_xpos(musc_ctx* c, int wide) {
  if (c->rdx_epoch == c->data_epoch && c->rdx_wide == wide) return true;
  if (ensure(c, c->rdx, c->nreads)) return false;
  if (wide)
    hipLaunchKernelGGL(k_read_xpos<true>, dim3(nblk(c->nreads, 256)), dim3(256), 0, c->stream, c->rd, c->rdm, c->nreads, c->rw, c->rdx.p);
  else
    hipLaunchKernelGGL(k_read_xpos<false>, dim3(nblk(c->nreads, 256)), dim3(256), 0, c->stream, c->rd, c->rdm, c->nreads, c->rw, c->rdx.p);
  if (hipGetLastError() != hipSuccess) return false;
  c->rdx_epoch = c->data_epoch;
  c->rdx_wide = wide;
  c->xok_epoch = ~0ull;
  c->xokdb_epoch = ~0ull;
  return true;
}

// Reads with X fit the context path if every read that holds more of them than its xpos word lists
// (XPos<wide>: four on 120-base buckets, three on wide ones) could not match anyway (that many
// mismatches exceed its budget int((1 - PMatch) * len)).  One small kernel and a 4-byte readback
// per (read set, bucket width, PMatch, MaxMismatch).
static bool reads_x_fit(musc_ctx* c, const musc_params* P, uint32_t max_len, int wide) {
  if (!c->rdm || !c->rd || !c->nreads) return false;
  if (c->env.no_x_context) return false;
  if (!reads_xpos(c, wide)) return false;
  // (the budget table covers the reads in hand whatever length the caller planned the index for)
  max_len = std::max(max_len, c->max_len);
  if (c->xok_epoch == c->data_epoch && c->xok_pmatch == P->pmatch && c->xok_mmp1 == P->max_mismatch_p1) return c->xok;
  std::vector<uint16_t> tab((size_t)max_len + 2);
  for (uint32_t L = 0; L < tab.size(); L++) {  // the budget exactly as musc_match_device builds it
    volatile double a = 1.0 - P->pmatch;
    volatile double b = a * (double)L;
    long long v = (long long)b;
    if (P->max_mismatch_p1 > 0) v = P->max_mismatch_p1 - 1;
    if (v < 0) v = 0;
    if (v > 0xFFFE) v = 0xFFFE;
    tab[L] = (uint16_t)v;
  }
  TmpBufs B;
  uint16_t* d_tab = nullptr;
  uint32_t bad = 1;
  if (B.alloc(&d_tab, tab.size() * 2) != hipSuccess) return false;
  if (hipMemcpyAsync(d_tab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice, c->stream) != hipSuccess) return false;
  if (hipMemsetAsync(c->d_flag, 0, 4, c->stream) != hipSuccess) return false;
  hipLaunchKernelGGL(k_xpos_check, dim3(nblk(c->nreads, 256)), dim3(256), 0, c->stream, c->rd, c->rdx.p, c->nreads, c->rw, d_tab,
                     max_len, wide ? XPos<true>::MAX : XPos<false>::MAX, c->d_flag);
  if (hipMemcpyAsync(&bad, c->d_flag, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return false;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return false;
  c->xok = bad == 0;
  c->xok_epoch = c->data_epoch;
  c->xok_pmatch = P->pmatch;
  c->xok_mmp1 = P->max_mismatch_p1;
  return c->xok;
}

// Reads with X against a DATABASE with X fit the context path if every read lists all its X in its
// xpos word and none of them falls into one of the run's windows (k_xpos_check_db).
static bool reads_x_fit_db(musc_ctx* c, const musc_params* P, int wide) {
  if (!c->rdm || !c->rd || !c->nreads) return false;
  if (!reads_xpos(c, wide)) return false;
  int32_t key[CTX_MAX_W + 3] = {P->n_windows, P->window_width, wide};
  XWins wn;
  memset(&wn, 0, sizeof wn);
  wn.n = P->n_windows;
  wn.ww = P->window_width;
  for (int k = 0; k < P->n_windows && k < CTX_MAX_W; k++) key[3 + k] = wn.q1[k] = P->windows[k];
  if (c->xokdb_epoch == c->data_epoch && memcmp(key, c->xokdb_key, sizeof key) == 0) return c->xokdb;
  uint32_t bad = 1;
  if (hipMemsetAsync(c->d_flag, 0, 4, c->stream) != hipSuccess) return false;
  if (wide) hipLaunchKernelGGL(k_xpos_check_db<true>, dim3(nblk(c->nreads, 256)), dim3(256), 0, c->stream, c->rdx.p, c->nreads, wn, c->d_flag);
  else hipLaunchKernelGGL(k_xpos_check_db<false>, dim3(nblk(c->nreads, 256)), dim3(256), 0, c->stream, c->rdx.p, c->nreads, wn, c->d_flag);
  if (hipMemcpyAsync(&bad, c->d_flag, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return false;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return false;
  c->xokdb = bad == 0;
  c->xokdb_epoch = c->data_epoch;
  memcpy(c->xokdb_key, key, sizeof key);
  return c->xokdb;
}

// Which index a run with these parameters and reads of at most max_len bases uses: context
// buckets when every read fits the context around each of at most CTX_MAX_W windows -- 120 bases
// (three entries per bucket line) or, where k_match_t runs, 200 bases (two per line: *wide = 1) --
// the database holds no X (the context has no mask plane; reads may hold some where
// k_match_t runs, see reads_x_fit) and positions fit 32 bits.
static bool ctx_eligible(musc_ctx* c, const musc_params* P, uint32_t max_len, int* CL, int* wide) {
  if (c->env.index != EnvKnobs::IDX_AUTO) return false;  // the two-kernel path
  // (the planes themselves may exist without an X: an all-zero one is made for the side that has
  // none when the other side does, and the database's stays for the context's lifetime)
  const bool lane = true;  // (both fused kernels read flagged entries, xpos words and wide buckets' runs: k_match_g only takes runs without them)
  // a database with X: k_match_t only (an entry whose context holds an X is flagged in bit 31 of its
  // position, the X's place or "several: see the mask plane" in the top byte of its target number)
  if (c->db_has_x && (!lane || c->max_tlen >= 0x80000000ull || c->nseq > (1u << 24) || c->env.no_x_context)) return false;
  if (c->nbases >= 0xFFFFFFF0ull || c->env.force_wide) return false;
  if (P->n_windows > CTX_MAX_W) return false;
  int q1min = P->windows[0], q1max = P->windows[0];
  for (int k = 1; k < P->n_windows; k++) {
    q1min = std::min(q1min, P->windows[k]);
    q1max = std::max(q1max, P->windows[k]);
  }
  const int64_t span = (int64_t)q1max - q1min + (int64_t)max_len;
  const int wenv = c->env.context;  // experiments: "narrow" (1) keeps runs beyond 120 bases on the two-kernel path, "wide" (2) puts every run on wide buckets
  *CL = q1max;
  *wide = 0;
  if (span > CTX_BASES || q1max > CTX_BASES || (wenv == 2 && lane)) {
    // wide buckets: k_match_t only; records of up to 16 words hold 200-base reads and their length word
    if (!lane || wenv == 1) return false;
    if (span > CTXW_BASES || q1max > CTXW_BASES) return false;
    *wide = 1;
  }
  // reads with X: k_match_t handles them, and only while every read either lists all its X
  // in its xpos word or has more X than mismatches allowed (reads_x_fit, cached per reads + budget)
  if (c->reads_have_x && !(lane && (c->db_has_x ? reads_x_fit_db(c, P, *wide) : reads_x_fit(c, P, max_len, *wide)))) return false;
  return true;
}

static int ensure_index(musc_ctx* c, const musc_params* P, uint32_t max_len) {
  int CL = 0, wide = 0;
  if (ctx_eligible(c, P, max_len, &CL, &wide)) {
    const int rc = build_index_ctx(c, P->window_width, CL, wide);
    if (rc != 100) return rc;
    // (does not fit the free memory: the classic index is a quarter of the size)
  }
  return musc_db_build_index(c, P->window_width);
}

int musc_db_build_index_for(musc_ctx* c, const musc_params* P, int32_t max_read_len) {
  if (!c) return 1;
  int rc = check_params(c, P);
  if (rc) return rc;
  if (!c->db2) return fail(c, 4, "no database loaded");
  HIPCHK(c, hipSetDevice(c->device));
  const uint32_t ml = max_read_len > 0 ? (uint32_t)max_read_len
                                       : (P->max_read_length > 0 ? (uint32_t)P->max_read_length : c->max_len);
  return ensure_index(c, P, ml);
}

// ---------------------------------------------------------------- reads

// The records of reads [r0, r0 + n) must exist before stream `st` goes on: for the pieces of an
// asynchronous upload (reads_load_fixed) that are still missing, `st` waits for their arrival and
// packs them.  A no-op once every record exists.
static int upload_prepare(musc_ctx* c, uint64_t r0, uint64_t n, hipStream_t st) {
  musc_ctx::Upload& u = c->up;
  if (!u.active) return 0;
  const uint64_t want = std::min<uint64_t>(r0 + n, c->nreads);
  while (u.packed_upto < want) {
    const uint64_t i = u.packed_upto / u.piece;
    const uint64_t first = i * u.piece, cnt = std::min<uint64_t>(u.piece, c->nreads - first);
    HIPCHK(c, hipStreamWaitEvent(st, u.ev[i], 0));
    hipLaunchKernelGGL(k_pack_reads_fixed, dim3(nblk(cnt * (uint64_t)c->rw, 256)), dim3(256), 0, st, u.stage, first, cnt, u.L, c->rw, c->rd);
    HIPCHK(c, hipGetLastError());
    u.packed_upto = first + cnt;
  }
  if (u.packed_upto >= c->nreads) u.active = false;
  return 0;
}

static int reads_load(musc_ctx* c, const unsigned char* ascii, const uint8_t* bases2bit, const uint8_t* nmask,
                      const uint64_t* offsets, uint64_t nreads, int on_device, bool packed) {
  HIPCHK(c, hipSetDevice(c->device));
  free_reads(c);
  if (nreads >= 0xFFFFFFF0ull) return fail(c, 2, "too many reads for 32-bit read_idx");
  c->nreads = nreads;
  if (nreads == 0) {
    c->rw = 4;
    return 0;
  }
  TmpBufs B;
  uint64_t* d_off = nullptr;
  const uint64_t* offp = offsets;
  if (!on_device) {
    HIPCHK(c, B.alloc(&d_off, (nreads + 1) * 8));
    HIPCHK(c, hipMemcpyAsync(d_off, offsets, (nreads + 1) * 8, hipMemcpyHostToDevice, c->stream));
    offp = d_off;
  }
  HIPCHK(c, hipMemsetAsync(c->counters + 4, 0, 8, c->stream));
  hipLaunchKernelGGL(k_max_len, dim3(std::min(nblk(nreads, 256), MAX_GRID)), dim3(256), 0, c->stream, offp, nreads, c->counters + 4);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->counters + 4, 8, hipMemcpyDeviceToHost, c->stream));
  uint64_t first = 0, total = 0;
  HIPCHK(c, hipMemcpyAsync(&first, offp, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&total, offp + nreads, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint64_t maxlen = c->h_pinned[0];
  if (first != 0) return fail(c, 2, "read offsets[0] must be 0");
  if (maxlen > 65535) return fail(c, 2, "read of %llu bases exceeds the 65535-base record limit", (unsigned long long)maxlen);
  c->max_len = (uint32_t)maxlen;
  int rw = (int)((2 * maxlen + 31) / 32) + 1;
  rw = (rw + 3) & ~3;
  if (rw < 4) rw = 4;
  c->rw = rw;
  const uint64_t words = nreads * (uint64_t)rw;
  if (words >= (1ull << 32)) return fail(c, 2, "too many read words for one dispatch (reads x record words >= 2^32)");
  HIPCHK(c, hipMalloc((void**)&c->rd, words * 4 + 256));
  HIPCHK(c, hipMemsetAsync(c->rd + words, 0, 256, c->stream));
  // the mask plane only where an X can turn up: ASCII input, or packed input that comes with a mask
  const bool may_have_x = !packed || nmask != nullptr;
  if (may_have_x) {
    HIPCHK(c, hipMalloc((void**)&c->rdm, words * 4 + 256));
    HIPCHK(c, hipMemsetAsync(c->rdm + words, 0, 256, c->stream));
  }
  uint32_t* d_hasx = c->d_flag;
  HIPCHK(c, hipMemsetAsync(d_hasx, 0, 4, c->stream));
  unsigned char *t1 = nullptr, *t2 = nullptr;
  if (!packed) {
    const unsigned char* src = ascii;
    if (!on_device && total) {
      HIPCHK(c, B.alloc(&t1, total));
      HIPCHK(c, hipMemcpyAsync(t1, ascii, total, hipMemcpyHostToDevice, c->stream));
      src = t1;
    }
    hipLaunchKernelGGL(k_pack_reads<false>, dim3(nblk(words, 256)), dim3(256), 0, c->stream, src,
                       (const uint32_t*)nullptr, (const uint32_t*)nullptr, offp, nreads, rw, c->rd, c->rdm, d_hasx);
  } else {
    const uint64_t b2 = ((total + 3) / 4 + 7) & ~3ull, bm = ((total + 7) / 8 + 7) & ~3ull;
    HIPCHK(c, B.alloc(&t1, b2 + 16));
    HIPCHK(c, hipMemsetAsync(t1, 0, b2 + 16, c->stream));
    HIPCHK(c, hipMemcpyAsync(t1, bases2bit, (total + 3) / 4, hipMemcpyHostToDevice, c->stream));
    if (nmask) {
      HIPCHK(c, B.alloc(&t2, bm + 16));
      HIPCHK(c, hipMemsetAsync(t2, 0, bm + 16, c->stream));
      HIPCHK(c, hipMemcpyAsync(t2, nmask, (total + 7) / 8, hipMemcpyHostToDevice, c->stream));
    }
    hipLaunchKernelGGL(k_pack_reads<true>, dim3(nblk(words, 256)), dim3(256), 0, c->stream,
                       (const unsigned char*)nullptr, (const uint32_t*)t1, (const uint32_t*)t2, offp, nreads, rw,
                       c->rd, c->rdm, d_hasx);
  }
  HIPCHK(c, hipGetLastError());
  uint32_t hasx = 0;
  HIPCHK(c, hipMemcpyAsync(&hasx, d_hasx, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->reads_have_x = hasx != 0;
  if (!hasx && c->rdm) {
    (void)hipFree(c->rdm);
    c->rdm = nullptr;
  }
  return 0;
}

// Reads of one length without X: the 2-bit stream is all that crosses PCIe.  async: the upload is
// queued in pieces on a copy stream and the records are made batch by batch by the pass that needs
// them (upload_prepare), so that upload and matching overlap.
static int reads_load_fixed(musc_ctx* c, const uint8_t* bases2bit, uint32_t L, uint64_t nreads, int async) {
  HIPCHK(c, hipSetDevice(c->device));
  free_reads(c);
  if (nreads >= 0xFFFFFFF0ull) return fail(c, 2, "too many reads for 32-bit read_idx");
  if (L > 65535) return fail(c, 2, "read of %u bases exceeds the 65535-base record limit", L);
  c->nreads = nreads;
  if (nreads == 0) {
    c->rw = 4;
    return 0;
  }
  c->max_len = L;
  int rw = (int)((2ull * L + 31) / 32) + 1;
  rw = (rw + 3) & ~3;
  if (rw < 4) rw = 4;
  c->rw = rw;
  const uint64_t words = nreads * (uint64_t)rw;
  if (words >= (1ull << 32)) return fail(c, 2, "too many read words for one dispatch (reads x record words >= 2^32)");
  HIPCHK(c, hipMalloc((void**)&c->rd, words * 4 + 256));
  HIPCHK(c, hipMemsetAsync(c->rd + words, 0, 256, c->stream));
  c->reads_have_x = false;
  musc_ctx::Upload& u = c->up;
  const uint64_t total_bytes = (nreads * (uint64_t)L + 3) / 4;
  const uint64_t need_words = (total_bytes + 3) / 4 + 4;  // (k_pack_reads_fixed reads up to two words past a read's last)
  if (u.stage_words < need_words) {
    if (u.stage) (void)hipFree(u.stage);
    u.stage = nullptr;
    u.stage_words = 0;
    HIPCHK(c, hipMalloc((void**)&u.stage, need_words * 4));
    u.stage_words = need_words;
  }
  if (!u.s_up) HIPCHK(c, hipStreamCreateWithFlags(&u.s_up, hipStreamNonBlocking));
  // pieces of whole bytes of the stream (4 | piece) and whole wave-tiles (64 | piece), a quarter of a
  // pass's batch each: the first batch can start when a quarter of it has arrived... no: a batch
  // needs all its pieces, but the NEXT batch's pieces arrive while this one is matched
  uint64_t piece = std::max<uint64_t>(c->batch_reads / 4, 64);
  piece = (piece + 63) & ~63ull;
  u.piece = piece;
  u.n_pieces = (nreads + piece - 1) / piece;
  u.L = L;
  u.packed_upto = 0;
  while (u.ev.size() < u.n_pieces) {
    hipEvent_t e = nullptr;
    HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    u.ev.push_back(e);
  }
  HIPCHK(c, hipMemsetAsync(u.stage + (need_words - 4), 0, 16, u.s_up));
  for (uint64_t i = 0; i < u.n_pieces; i++) {
    const uint64_t b0 = i * piece * (uint64_t)L / 4;  // (64 | piece: a whole number of bytes)
    const uint64_t b1 = std::min<uint64_t>(((i + 1) * piece * (uint64_t)L + 3) / 4, total_bytes);
    HIPCHK(c, hipMemcpyAsync(reinterpret_cast<uint8_t*>(u.stage) + b0, bases2bit + b0, b1 - b0, hipMemcpyHostToDevice, u.s_up));
    HIPCHK(c, hipEventRecord(u.ev[i], u.s_up));
  }
  u.active = true;
  if (!async) {
    int rc = upload_prepare(c, 0, nreads, c->stream);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return 0;
}

int musc_reads_load_packed32(musc_ctx* c, const uint8_t* bases2bit, const uint8_t* nmask, const uint32_t* lengths,
                             uint32_t fixed_len, uint64_t nreads, int async) {
  if (!c) return 1;
  if (!bases2bit && nreads) return fail(c, 2, "musc_reads_load_packed32: NULL input");
  if (!lengths && !nmask) return reads_load_fixed(c, bases2bit, fixed_len, nreads, async);
  // lengths and / or a mask: offsets are made on the device (a scan of the lengths, or multiples of
  // fixed_len) and the general loader takes over
  HIPCHK(c, hipSetDevice(c->device));
  TmpBufs B;
  uint64_t *d_off = nullptr, *stmp = nullptr;
  uint32_t* d_len = nullptr;
  HIPCHK(c, B.alloc(&d_off, (nreads + 2) * 8));
  if (lengths) {
    HIPCHK(c, B.alloc(&d_len, (nreads + 1) * 4));
    HIPCHK(c, B.alloc(&stmp, scan_tmp_elems(nreads + 1) * 8));
    HIPCHK(c, hipMemcpyAsync(d_len, lengths, nreads * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_widen_u32, dim3(nblk(nreads + 1, 256)), dim3(256), 0, c->stream, d_len, nreads, d_off);
    HIPCHK(c, hipGetLastError());
    int rc = scan_u64(c, d_off, d_off, nreads + 1, stmp);
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL(k_iota_mul, dim3(nblk(nreads + 1, 256)), dim3(256), 0, c->stream, d_off, nreads, (uint64_t)fixed_len);
    HIPCHK(c, hipGetLastError());
  }
  return reads_load(c, nullptr, bases2bit, nmask, d_off, nreads, 1, true);
}

int musc_reads_load_ascii(musc_ctx* c, const char* seqs, const uint64_t* offsets, uint64_t nreads, int on_device) {
  if (!c) return 1;
  if ((!seqs || !offsets) && nreads) return fail(c, 2, "musc_reads_load_ascii: NULL input");
  return reads_load(c, (const unsigned char*)seqs, nullptr, nullptr, offsets, nreads, on_device, false);
}

int musc_reads_load_packed(musc_ctx* c, const uint8_t* bases2bit, const uint8_t* nmask, const uint64_t* read_offsets,
                           uint64_t nreads) {
  if (!c) return 1;
  if ((!bases2bit || !read_offsets) && nreads) return fail(c, 2, "musc_reads_load_packed: NULL input");
  return reads_load(c, nullptr, bases2bit, nmask, read_offsets, nreads, 0, true);
}

// ---------------------------------------------------------------- hot path

// One pass on context buckets: per batch k_match (screen + confirm + select, tuples staged per
// workgroup) -> scan of the per-tile tuple counts -> k_compact.  The only data-dependent
// capacities are the staging region and the spill region of a workgroup; the first pass over a
// (reads, database, parameters) combination sizes them (a batch that does not fit makes the pass
// start over with larger buffers), later passes run without host round trips and check the
// guards once at the end -- the same protocol as the two-kernel path below.
extern "C++" {
static size_t match_dyn_lds(int kind, int W, int block_mode) {
  // per-(window, read) counters of the wave-tile in hand (k_match_t: of two wave-tiles), then (mode 1) the sketch
  if (kind == MK_DMA) return block_mode == 1 ? (size_t)(4u << MATCHG_SKETCH_BITS) : 0u;  // (its per-(window, read) counters are registers)
  const size_t wcnt = (size_t)TILE * W * 4 * (kind == MK_LANE ? 2 : 1);  // TILE = 4 waves x 64
  return block_mode ? wcnt + (block_mode == 1 ? (4u << MATCH_SKETCH_BITS) : 0u) : 0u;
}

// the kernel a pass launches, as a function pointer (occupancy queries and attributes)
template <int RW>
static const void* match_fn(const musc_ctx* c, int W) {
  const int kind = match_kind(c, W);
  const bool rx = c->reads_have_x;
  if (kind == MK_DMA) {
    if constexpr (RW == 8) return c->spec_geom == 1 ? reinterpret_cast<const void*>(&k_match_g<8, 1>) : reinterpret_cast<const void*>(&k_match_g<8, 0>);
    else return nullptr;
  }
  if (kind == MK_LANE) {
    if constexpr (RW == 8) {
      if (c->spec_geom == 1) return reinterpret_cast<const void*>(&k_match_t<8, 2, 0, false, 1>);
    }
    // (instances: 120-base buckets for records of 4, 8, 12 words; wide ones for 4 to 16)
#define MUSC_LANE_FN2(WN, WD) (c->db_has_x ? reinterpret_cast<const void*>(&k_match_t<RW, WN, 2, WD, 0>) : rx ? reinterpret_cast<const void*>(&k_match_t<RW, WN, 1, WD, 0>) : reinterpret_cast<const void*>(&k_match_t<RW, WN, 0, WD, 0>))
#define MUSC_LANE_FN(WN)                                  \
  if (c->idx_wide) return MUSC_LANE_FN2(WN, true);        \
  if constexpr (RW <= 12) return MUSC_LANE_FN2(WN, false); \
  return nullptr;
    switch (W) {
      case 1: MUSC_LANE_FN(1)
      case 2: MUSC_LANE_FN(2)
      case 3: MUSC_LANE_FN(3)
      default: MUSC_LANE_FN(4)
    }
#undef MUSC_LANE_FN
#undef MUSC_LANE_FN2
  }
  return nullptr;
}

// workgroups of the kernel that are resident at once on this device: the persistent grid
template <int RW>
static unsigned match_resident(musc_ctx* c, int W, int block_mode) {
  int per_cu = 0, ncu = 0;
  const size_t lds = match_dyn_lds(match_kind(c, W), W, block_mode);
  const void* fn = match_fn<RW>(c, W);
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, TILE, lds);
  if (e != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 2; }
  // The occupancy query counts LDS to the byte; the hardware hands it out in larger pieces
  // (measured on gfx950: 3 x 54 208 B did not fit a CU's 160 KB, 3 x 52 160 B did), and a grid one
  // workgroup per CU too large runs its last third as a second round (+45 % on cfg3).  Bound the
  // count with 2 KB pieces.
  {
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, fn) == hipSuccess) {
      const size_t total = ((size_t)fa.sharedSizeBytes + lds + 2047) / 2048 * 2048;
      const int fit = total ? (int)((160u << 10) / total) : per_cu;
      if (fit >= 1 && fit < per_cu) per_cu = fit;
    } else {
      (void)hipGetLastError();
    }
  }
  if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || ncu < 1) ncu = 256;
  unsigned resident = (unsigned)per_cu * (unsigned)ncu;
  if (c->env.debug_grid >= 1 && (unsigned)c->env.debug_grid < resident) resident = (unsigned)c->env.debug_grid;  // tests: a small grid makes every wave walk many wave-tiles of a small input
  return resident;
}

// set: which staging buffers the launch fills (0: stage / tcount2, 1: stage_b / tcount2_b);
// prev_tiles > 0: the launch also moves the other set's tuples (the previous batch's) into `hits`
template <int RW>
static void launch_match(musc_ctx* c, uint64_t r0, uint32_t n, int W, int block_mode, uint32_t block_thr,
                         unsigned ngrid, int set = 0, uint32_t prev_tiles = 0) {
  const dim3 grid(ngrid), block(TILE);
  const int kind = match_kind(c, W);
  const size_t lds = match_dyn_lds(kind, W, block_mode);
  DevBuf<uint4>& st = set ? c->stage_b : c->stage;
  DevBuf<uint32_t>& tc = set ? c->tcount2_b : c->tcount2;
#define MUSC_LAUNCH_MATCH(K, ...)                                                                                 \
  hipLaunchKernelGGL(K, grid, block, lds, c->stream, c->rd, r0, n, c->d_mp, c->nmiss_tab.p,                       \
                     c->ctx_T, c->ctx_E, st.p, c->stage.cap, c->spill.p, c->spill.cap, c->bs[0].tbase.p,          \
                     tc.p, block_mode, block_thr, c->block_table.p, c->counters, ##__VA_ARGS__)
  {
    const uint4* pst = prev_tiles ? (set ? c->stage.p : c->stage_b.p) : nullptr;
    const uint32_t* ptc = set ? c->tcount2.p : c->tcount2_b.p;
    const uint32_t* ptp = set ? c->tpre.p : c->tpre_b.p;
    uint4* const hp = reinterpret_cast<uint4*>(c->hits.p);
    const uint32_t* const rdx = c->reads_have_x ? (const uint32_t*)c->rdx.p : (const uint32_t*)nullptr;
    if (kind == MK_DMA) {
      if constexpr (RW == 8) {
        if (c->spec_geom == 1) MUSC_LAUNCH_MATCH((k_match_g<8, 1>), pst, ptc, ptp, prev_tiles, hp, c->hits.cap, rdx);
        else MUSC_LAUNCH_MATCH((k_match_g<8, 0>), pst, ptc, ptp, prev_tiles, hp, c->hits.cap, rdx);
      }
      return;
    }
    if (kind == MK_LANE) {
      if constexpr (RW == 8) {
        if (c->spec_geom == 1) {  // (chosen by spec_geom_matches: every specialised quantity equals the run's)
          MUSC_LAUNCH_MATCH((k_match_t<8, 2, 0, false, 1>), pst, ptc, ptp, prev_tiles, hp, c->hits.cap, rdx);
          return;
        }
      }
#define MUSC_LAUNCH_LANE2(WN, WD)                                                                                 \
      if (c->db_has_x) MUSC_LAUNCH_MATCH((k_match_t<RW, WN, 2, WD, 0>), pst, ptc, ptp, prev_tiles, hp, c->hits.cap, rdx); \
      else if (c->reads_have_x) MUSC_LAUNCH_MATCH((k_match_t<RW, WN, 1, WD, 0>), pst, ptc, ptp, prev_tiles, hp, c->hits.cap, rdx); \
      else MUSC_LAUNCH_MATCH((k_match_t<RW, WN, 0, WD, 0>), pst, ptc, ptp, prev_tiles, hp, c->hits.cap, rdx)
#define MUSC_LAUNCH_LANE(WN)                                      \
      do {                                                        \
        if (c->idx_wide) { MUSC_LAUNCH_LANE2(WN, true); }         \
        else if constexpr (RW <= 12) { MUSC_LAUNCH_LANE2(WN, false); } \
      } while (0)
      switch (W) {
        case 1: MUSC_LAUNCH_LANE(1); break;
        case 2: MUSC_LAUNCH_LANE(2); break;
        case 3: MUSC_LAUNCH_LANE(3); break;
        default: MUSC_LAUNCH_LANE(4); break;
      }
#undef MUSC_LAUNCH_LANE2
#undef MUSC_LAUNCH_LANE
      return;
    }
  }
#undef MUSC_LAUNCH_MATCH
}
}  // extern "C++"

static int match_device_impl(musc_ctx* c, const musc_params* P, uint64_t* nhits);

// The geometry-specialised instance a pass may launch: SpecGeom<g> is taken only when EVERY quantity it turns into
// a constant equals the run's -- window width, window starts, context offset, MinDinuc, the first-window sets, and
// the TABLE: a direct table of 2 * ww bits (cfg2's 10^8-base database gets a hashed 2^27-bucket table for the same
// ww: the general instance) -- and the instance exists for this record stride / bucket width / X mode.  Reads of
// other lengths than the geometry's are fine: the instance falls back to per-lane length masks for such a tile.
// MUSC_NO_SPEC=1 keeps every pass on the general instances (A/B runs, tests).
extern "C++" {
template <int SG>
static bool spec_geom_equals(const MatchParams& mp) {
  typedef SpecGeom<SG> G;
  if (mp.W != G::nwin || mp.ww != G::ww || mp.CL != G::CL || mp.min_dinuc != G::min_dinuc || mp.direct != 1 || mp.bits != 2 * G::ww) return false;
  for (int k = 0; k < G::nwin; k++)
    if (mp.win[k] != G::win[k] || mp.need[k] != (1u << k) - 1u) return false;
  return true;
}
}  // extern "C++"
static int spec_geom_matches(const musc_ctx* c, const MatchParams& mp) {
  if (c->env.no_spec) return 0;
  if (c->rw != 8 || c->idx_wide || c->db_has_x || c->reads_have_x) return 0;  // the instances that exist: <8, 2, 0, false, g>
  if (spec_geom_equals<1>(mp)) return 1;
  return 0;
}

static int match_ctx_pass(musc_ctx* c, const musc_params* P, const PathParams& pp, int block_mode, uint32_t block_thr_unused,
                          uint64_t max_matches, uint64_t planned_batches, uint64_t* nhits) {
  int rc = 0;
  if (c->rw != 4 && c->rw != 8 && c->rw != 12 && !(c->rw == 16 && c->idx_wide))
    return fail(c, 12, "internal: record stride %d on the context path", c->rw);
  (void)block_thr_unused;
  // the run's parameter block (and with it the kernel instance: match_fn looks at c->spec_geom)
  {
    static thread_local MatchParams mp;  // 16 KB with its mask tables: not on the stack
    memset(&mp, 0, sizeof mp);
    mp.W = pp.W; mp.ww = pp.ww; mp.min_dinuc = pp.min_dinuc; mp.bits = pp.bits; mp.direct = pp.direct;
    mp.mmtol = pp.mmtol; mp.apply_mmtol = pp.apply_mmtol; mp.max_len = pp.max_len; mp.CL = c->idx_CL;
    mp.q1zero_mask = pp.q1zero_mask;
    mp.seq_off = c->db_has_x ? c->seq_off : nullptr;
    mp.dbm2 = c->db_has_x ? c->dbm2 : nullptr;
    for (int k = 0; k < pp.W && k < CTX_MAX_W; k++) mp.win[k] = pp.win[k];
    match_tables(mp);
    c->spec_geom = spec_geom_matches(c, mp);
    c->stats.match_variant = (match_kind(c, pp.W) == MK_DMA ? 4u : 2u) + (c->spec_geom ? 1u : 0u);
    if (!c->h_mp_valid || memcmp(&mp, &c->h_mp, sizeof mp) != 0) {
      c->h_mp = mp;
      HIPCHK(c, hipMemcpyAsync(c->d_mp, &c->h_mp, sizeof mp, hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      c->h_mp_valid = true;
    }
  }
  // the persistent grid = the workgroups that are resident at once (every wave then sees many
  // wave-tiles and the end-of-kernel atomics stay few); the MaxMatches screening threshold is per
  // workgroup-launch, so it follows the grid
  const unsigned resident1 = c->rw == 4 ? match_resident<4>(c, pp.W, block_mode)
                             : c->rw == 8 ? match_resident<8>(c, pp.W, block_mode)
                             : c->rw == 12 ? match_resident<12>(c, pp.W, block_mode)
                                           : match_resident<16>(c, pp.W, block_mode);
  // MUSC_GRID_ROUNDS (an experiment knob; default 1): a grid of that many times the resident workgroups.  The waves
  // of a launch do not run at one speed (r04: the slowest wave of a cfg3 launch takes 1.2-1.3 x the mean), which
  // looks like a tail a dynamically dispatched grid would remove -- it does not: 2 / 4 / 8 / 16 rounds run the cfg3
  // launch in 1.03 / 1.04 / 1.06 / 1.11 ms against 0.98-0.99 (profiles/r04_ab_shape_spec_dma.txt).  The memory system
  // is the shared resource: the waves that finish early leave their share to the slow ones, nothing idles.
  const unsigned rounds = c->env.debug_grid >= 1 ? 1u : (c->env.grid_rounds >= 1 && c->env.grid_rounds <= 64 ? (unsigned)c->env.grid_rounds : (unsigned)MATCH_GRID_ROUNDS);
  const unsigned resident = resident1 * rounds;
  uint32_t block_thr = (uint32_t)std::min<uint64_t>(max_matches / (planned_batches * resident), 0x7FFFFFFFull);
  if (block_mode == 1 && block_thr < 2) block_mode = 2;
  if (block_mode == 2 && !c->block_table.p) {
    if ((rc = ensure(c, c->block_table, 1ull << BLOCK_TABLE_BITS))) return rc;
  }
  const bool check_blocks = block_mode != 0;
  const bool sized = c->sized_epoch == c->data_epoch && c->sized_exact_blocks == (block_mode == 2) &&
                     memcmp(&c->sized_params, P, sizeof *P) == 0 && !c->env.debug_sync;
  uint32_t bsz = sized ? c->sized_bsz : c->batch_reads;
  const uint64_t L = c->max_len;
  // A sized pass can be replayed as a hipGraph (MUSC_GRAPH=1): its launches, the counter memsets
  // and the final readback are captured once per (reads, database, parameters) and then cost one
  // launch per pass.  Every buffer of a sized pass is fixed, so the captured arguments stay valid;
  // any pass that sizes drops the graph.
  const bool use_graph = sized && c->env.graph > 0 && !c->graph_failed;
  if (!sized && c->graph_exec) {
    (void)hipGraphExecDestroy(c->graph_exec);
    c->graph_exec = nullptr;
  }
  // Where a batch's staged tuples go to their place in `hits`: a k_compact_w per batch (the default since r04), or
  // -- MUSC_FUSED_COMPACT=1, r02 / r03's default -- from inside the NEXT batch's match launch.  Measured on cfg3 / the cfg4
  // shard (profiles/r04_ab_shape_spec_dma.txt): the pass takes the same time either way (3.09-3.24 against 3.25 ms; 1.78
  // against 1.74), but a launch that also moves 12 M tuples takes 0.96-1.00 ms instead of 0.94-0.95 for work its
  // algorithmic bytes do not bill: the separate kernel keeps the match kernel's accounting exact.
  const bool fuse_ok = c->env.fused_compact;
  if (sized && fuse_ok && c->nreads > bsz) {  // the second staging set (allocated outside any capture)
    if ((rc = ensure(c, c->stage_b, c->stage.cap)) || (rc = ensure(c, c->tcount2_b, c->tcount2.cap)) ||
        (rc = ensure(c, c->tpre_b, c->tpre.cap)))
      return rc;
  }
  for (int attempt = 0;; attempt++) {
    if (attempt > 40) return fail(c, 12, "internal: the context pass did not converge on buffer sizes");
    const bool fuse = fuse_ok && c->nreads > bsz;  // (a sizing pass may have halved the batch size)
    Timer tm(c);
    hipEvent_t ev0 = pool_event(c), ev1 = pool_event(c);
    if (!ev0 || !ev1) return fail(c, 10, "hipEventCreate failed");
    const bool replay = use_graph && c->graph_exec && c->graph_epoch == c->data_epoch && c->graph_block_mode == block_mode &&
                        memcmp(&c->graph_params, P, sizeof *P) == 0;
    const bool capture = use_graph && !replay;
    if (capture && c->graph_exec) {
      (void)hipGraphExecDestroy(c->graph_exec);
      c->graph_exec = nullptr;
    }
    tm.off = capture || replay;
    CaptureGuard cap;  // (ends the capture if this attempt leaves early)
    if (!replay) {
      if (capture) {
        if (cap.begin(c->stream) != hipSuccess) {  // no capture on this stream: the plain sized pass
          (void)hipGetLastError();
          c->graph_failed = true;
          return match_device_impl(c, P, nhits);
        }
      } else {
        HIPCHK(c, hipEventRecord(ev0, c->stream));
      }
    }
    uint64_t n_cand = 0, n_cmp = 0, n_windows = 0, n_ovf = 0, r0 = 0;
    c->stats.n_batches = 0;
    c->stats.match_launches = 0;
    bool again = false;
    if (!replay) {
      HIPCHK(c, hipMemsetAsync(c->counters, 0, 16 * sizeof(unsigned long long), c->stream));
      if (block_mode == 2) HIPCHK(c, hipMemsetAsync(c->block_table.p, 0, (1ull << BLOCK_TABLE_BITS) * 4, c->stream));
    }
    int set = 0;
    uint32_t pending_tiles = 0;  // wave-tiles of the previous batch whose tuples are still staged
    while (!replay && r0 < c->nreads) {
      const uint32_t n = (uint32_t)std::min<uint64_t>(bsz, c->nreads - r0);
      const uint32_t ntiles = nblk(n, WT);  // wave-tiles of 64 reads
      const uint64_t sgrid = std::min<uint64_t>(nblk(n, TILE), resident);
      const uint64_t swaves = sgrid * (TILE / 64);  // regions of stage and spill are per wave
      if (!sized) {
        if ((rc = ensure(c, c->bs[0].tbase, (uint64_t)ntiles + 1))) return rc;
        if ((rc = ensure(c, c->scan_tmp, scan_tmp_elems((uint64_t)ntiles + 1)))) return rc;
        if ((rc = ensure(c, c->tcount2, (uint64_t)ntiles + 1))) return rc;
        if ((rc = ensure(c, c->tpre, (uint64_t)ntiles + 1))) return rc;
        if ((rc = ensure(c, c->stage, std::max<uint64_t>(2ull * n, swaves * 64)))) return rc;
        if ((rc = ensure(c, c->spill, swaves * 32))) return rc;
        if (fuse && ((rc = ensure(c, c->stage_b, c->stage.cap)) || (rc = ensure(c, c->tcount2_b, c->tcount2.cap)) ||
                     (rc = ensure(c, c->tpre_b, c->tpre.cap))))
          return rc;
        HIPCHK(c, hipMemsetAsync(c->counters + 8, 0, 8 * sizeof(unsigned long long), c->stream));
      }
      // A pass of k_match_t over several batches: the tuples batch b staged are moved into
      // `hits` by the launch of batch b + 1 (other staging set, same grid so the same regions);
      // only the last batch needs k_compact_w.
      const bool more = r0 + n < c->nreads;
      const uint64_t next_grid = more ? std::min<uint64_t>(nblk((uint32_t)std::min<uint64_t>(bsz, c->nreads - r0 - n), TILE), resident) : 0;
      const bool defer = fuse && more && next_grid == sgrid;  // this batch's tuples wait for the next launch
      if ((rc = upload_prepare(c, r0, n, c->stream))) return rc;  // (reads still on their way from the host)
      tm.begin(0);
      {
        Range rg("k_match");
      switch (c->rw) {
        case 4: launch_match<4>(c, r0, n, pp.W, block_mode, block_thr, (unsigned)sgrid, set, pending_tiles); break;
        case 8: launch_match<8>(c, r0, n, pp.W, block_mode, block_thr, (unsigned)sgrid, set, pending_tiles); break;
        case 12: launch_match<12>(c, r0, n, pp.W, block_mode, block_thr, (unsigned)sgrid, set, pending_tiles); break;
        default: launch_match<16>(c, r0, n, pp.W, block_mode, block_thr, (unsigned)sgrid, set, pending_tiles); break;
      }
      }
      HIPCHK(c, hipGetLastError());
      tm.end(0);
      if (pending_tiles) {  // the previous batch is in `hits` now
        hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, c->stream, set ? c->tpre.p : c->tpre_b.p, pending_tiles, c->counters);
        HIPCHK(c, hipGetLastError());
        pending_tiles = 0;
      }
      c->stats.match_launches++;
      c->stats.n_batches++;
      if (!sized) {
        HIPCHK(c, hipMemcpyAsync(&c->h_pinned[0], c->counters, 16 * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        // (k_match_t keeps a spill region per wave and wave-tile parity)
        const uint64_t need_stage = c->h_pinned[8 + 7] * swaves, need_spill = 2 * c->h_pinned[8 + 5] * swaves;
        if (need_stage > (1ull << 31)) {  // u32 tuple offsets within a batch: retry with half the reads
          if (n == 1) return fail(c, 6, "one read has %llu tuples (> 2^31)", (unsigned long long)c->h_pinned[8 + 7]);
          bsz = n / 2;
          again = true;
          break;
        }
        if (c->h_pinned[3] & 8ull)
          return fail(c, 12, "internal: the kernel instance specialised for geometry %d refused this run's parameters", c->spec_geom);
        if (c->h_pinned[3] || need_stage > c->stage.cap || need_spill > c->spill.cap) {
          // room for every workgroup's tuples / spilled candidates, then the pass starts over
          // (k_match has already added this batch to the pass-level counters)
          if (need_stage > c->stage.cap && (rc = ensure(c, c->stage, need_stage + need_stage / 4 + swaves))) return rc;
          if (need_spill > c->spill.cap && (rc = ensure(c, c->spill, need_spill + need_spill / 4 + swaves))) return rc;
          again = true;
          break;
        }
        n_windows += c->h_pinned[8 + 0];
        n_cmp += c->h_pinned[8 + 1];
        n_cand += c->h_pinned[8 + 3];
        n_ovf += c->h_pinned[8 + 4];
        if ((rc = ensure(c, c->hits, c->h_pinned[2] + c->h_pinned[8 + 6], true))) return rc;
      }
      Range rgc("scan + k_compact_w");
      tm.begin(4);
      tm.begin(1);
      rc = scan_u32(c, set ? c->tcount2_b.p : c->tcount2.p, set ? c->tpre_b.p : c->tpre.p, (uint64_t)ntiles + 1, false,
                    c->scan_tmp.p, c->stream);
      if (rc) return rc;
      tm.end(1);
      if (defer) {
        pending_tiles = ntiles;
        set ^= 1;
      } else {
        hipLaunchKernelGGL(k_compact_w, dim3(std::min(nblk(ntiles, 4), 4u * MAX_GRID)), dim3(256), 0, c->stream, ntiles,
                           c->bs[0].tbase.p, set ? c->tcount2_b.p : c->tcount2.p, set ? c->tpre_b.p : c->tpre.p,
                           set ? c->stage_b.p : c->stage.p, reinterpret_cast<uint4*>(c->hits.p), c->hits.cap, c->counters);
        HIPCHK(c, hipGetLastError());
        hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, c->stream, set ? c->tpre_b.p : c->tpre.p, ntiles, c->counters);
        HIPCHK(c, hipGetLastError());
      }
      tm.end(4);
      r0 += n;
    }
    if (again) continue;
    c->last_pp = pp;
    c->last_max_matches = (uint32_t)max_matches;
    c->last_exact_blocks = block_mode == 2;
    if (!replay) {
      if (block_mode == 2) {
        hipLaunchKernelGGL(k_block_overflow, dim3(1024), dim3(256), 0, c->stream, c->block_table.p, (uint32_t)max_matches,
                           c->counters);
        HIPCHK(c, hipGetLastError());
      }
      if (!capture) HIPCHK(c, hipEventRecord(ev1, c->stream));
      HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->counters, 16 * 8, hipMemcpyDeviceToHost, c->stream));
    }
    if (capture) {
      hipGraph_t g = nullptr;
      hipError_t ge = cap.end(&g);
      if (ge == hipSuccess) ge = hipGraphInstantiate(&c->graph_exec, g, nullptr, nullptr, 0);
      if (g) (void)hipGraphDestroy(g);
      if (ge != hipSuccess) {  // the graph is an optimisation: without it the pass runs launch by launch
        (void)hipGetLastError();
        c->graph_exec = nullptr;
        c->graph_failed = true;
        return match_device_impl(c, P, nhits);
      }
      c->graph_epoch = c->data_epoch;
      c->graph_params = *P;
      c->graph_block_mode = block_mode;
      c->graph_batches = c->stats.n_batches;
    }
    if (capture || replay) {
      c->stats.n_batches = c->stats.match_launches = c->graph_batches;
      HIPCHK(c, hipEventRecord(ev0, c->stream));
      HIPCHK(c, hipGraphLaunch(c->graph_exec, c->stream));
      HIPCHK(c, hipEventRecord(ev1, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->h_pinned[3] & 8ull)
      return fail(c, 12, "internal: the kernel instance specialised for geometry %d refused this run's parameters", c->spec_geom);
    if (c->h_pinned[3]) {
      if (!sized) return fail(c, 12, "internal: a capacity guard fired although every batch was sized (flags %llu)",
                              (unsigned long long)c->h_pinned[3]);
      c->sized_epoch = 0;  // the pass did not fit after all: run it the careful way
      return match_device_impl(c, P, nhits);
    }
    if (sized) {
      n_windows = c->h_pinned[8 + 0];
      n_cmp = c->h_pinned[8 + 1];
      n_cand = c->h_pinned[8 + 3];
      n_ovf = c->h_pinned[8 + 4];
    }
    c->stats.n_read_windows = n_windows;
    c->stats.n_candidates = n_cand;
    c->stats.n_pairs = n_cmp;
    c->stats.n_descriptors = 0;
    c->stats.n_overflow_entries = n_ovf;
    c->stats.n_accepted = c->h_pinned[1];
    c->stats.n_hits = c->nhits = c->h_pinned[2];
    c->stats.n_overflow_blocks = check_blocks ? c->h_pinned[5] : ~0ull;
    if (block_mode == 1 && (c->h_pinned[6] || c->stats.n_batches > planned_batches)) {
      c->force_exact_blocks = true;  // screening inconclusive: repeat with exact per-block counters
      c->exact_epoch = c->data_epoch;
      c->exact_params = *P;
      rc = match_device_impl(c, P, nhits);
      c->force_exact_blocks = false;
      return rc;
    }
    c->stats.ms_screen = tm.total(0);
    c->stats.ms_scan = tm.total(1);
    c->stats.ms_select = tm.total(4);
    (void)hipEventElapsedTime(&c->stats.ms_total, ev0, ev1);
    const uint64_t rec_b = (2 * L + 7) / 8;
    const uint64_t ent_b = c->idx_wide ? sizeof(CtxEntryW) : sizeof(CtxEntry);
    c->stats.match_bytes = c->nreads * rec_b + n_windows * sizeof(CtxBucket) + n_ovf * ent_b + 16 * c->stats.n_hits;
    c->stats.match_bytes_strict = c->nreads * rec_b + n_windows * 8 + n_cand * ent_b + 16 * c->stats.n_hits;
    if (nhits) *nhits = c->nhits;
    c->sized_epoch = c->data_epoch;
    c->sized_params = *P;
    c->sized_exact_blocks = block_mode == 2;
    c->sized_bsz = bsz;
    return 0;
  }
}

// musc_reads_load_packed32(async = 1) borrows the caller's host buffer "until the next musc_match* returns": that holds
// on every exit -- a pass that fails early (parameters, no database, an index that cannot be built, a HIP error in a
// batch) waits for the copies still queued on the upload stream before it returns.  The pieces stay valid on the
// device, so a later pass packs and matches them.
int musc_match_device(musc_ctx* c, const musc_params* P, uint64_t* nhits) {
  if (!c) return 1;
  const int rc = match_device_impl(c, P, nhits);
  if (rc != 0 && c->up.active && c->up.s_up) (void)hipStreamSynchronize(c->up.s_up);
  return rc;
}

static int match_device_impl(musc_ctx* c, const musc_params* P, uint64_t* nhits) {
  Range rg_pass("musc_match_device");
  int rc = check_params(c, P);
  if (rc) return rc;
  if (!c->db2) return fail(c, 4, "no database loaded");
  if (!c->rd && c->nreads) return fail(c, 4, "no reads loaded");
  HIPCHK(c, hipSetDevice(c->device));
  rc = ensure_index(c, P, c->max_len);
  if (rc) return rc;

  const float keep_index_ms = c->stats.ms_index_build;
  memset(&c->stats, 0, sizeof c->stats);
  c->stats.ms_index_build = keep_index_ms;
  c->stats.n_reads = c->nreads;
  c->nhits = 0;
  if (nhits) *nhits = 0;

  PathParams pp;
  memset(&pp, 0, sizeof pp);
  pp.W = P->n_windows;
  pp.ww = P->window_width;
  pp.min_dinuc = P->min_dinuc;
  pp.bits = c->idx_bits;
  pp.direct = c->idx_direct;
  pp.mmtol = P->mmtol > 0xFFFF ? 0xFFFF : P->mmtol;
  pp.apply_mmtol = P->apply_mmtol;
  pp.wide = c->wide;
  pp.max_len = (int32_t)c->max_len;
  for (int k = 0; k < pp.W; k++) {
    pp.win[k] = P->windows[k];
    if (P->windows[k] == 0) pp.q1zero_mask |= 1u << k;
  }

  // nmiss budget per read length: int((1-PMatch)*float64(len)), IEEE double, truncation
  // (cmd/muscato_confirm/main.go:198) -- evaluated on the host exactly as Go does.
  if (c->nm_pmatch != P->pmatch || c->nm_mmp1 != P->max_mismatch_p1 || c->nm_maxlen != c->max_len || !c->nmiss_tab.p) {
    std::vector<uint16_t> tab((size_t)c->max_len + 2);
    for (uint32_t L = 0; L < tab.size(); L++) {
      volatile double a = 1.0 - P->pmatch;
      volatile double b = a * (double)L;
      long long v = (long long)b;
      if (P->max_mismatch_p1 > 0) v = P->max_mismatch_p1 - 1;  // --MaxMismatch addition
      if (v < 0) v = 0;
      if (v > 0xFFFE) v = 0xFFFE;
      tab[L] = (uint16_t)v;
    }
    rc = ensure(c, c->nmiss_tab, tab.size());
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->nmiss_tab.p, tab.data(), tab.size() * 2, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // tab goes out of scope
    c->nm_pmatch = P->pmatch;
    c->nm_mmp1 = P->max_mismatch_p1;
    c->nm_maxlen = c->max_len;
  }

  HIPCHK(c, hipMemsetAsync(c->counters, 0, 8 * sizeof(unsigned long long), c->stream));
  // MaxMatches accounting (see k_confirm): screening first, exact only if inconclusive
  const uint64_t planned_batches = (c->nreads + c->batch_reads - 1) / c->batch_reads + 1;
  uint64_t max_matches = P->max_matches > 0 ? (uint64_t)P->max_matches : 0x7FFFFFFFull;
  if (P->n_shards > 1) max_matches /= (uint64_t)P->n_shards;  // this context sees one shard of each block
  const uint32_t block_thr = (uint32_t)std::min<uint64_t>(max_matches / (planned_batches * MAX_GRID), 0x7FFFFFFFull);
  const bool known_exact = c->exact_epoch == c->data_epoch && memcmp(&c->exact_params, P, sizeof *P) == 0;
  int block_mode = P->skip_block_check ? 0 : (c->force_exact_blocks || known_exact || block_thr < 2 ? 2 : 1);
  const bool check_blocks = block_mode != 0;
  c->cur_block_mode = block_mode;
  c->cur_block_thr = block_thr;
  if (block_mode == 2) {
    if ((rc = ensure(c, c->block_table, 1ull << BLOCK_TABLE_BITS))) return rc;
    HIPCHK(c, hipMemsetAsync(c->block_table.p, 0, (1ull << BLOCK_TABLE_BITS) * 4, c->stream));
  }
  if (!c->h_pp_valid || memcmp(&pp, &c->h_pp, sizeof pp) != 0) {
    c->h_pp = pp;
    HIPCHK(c, hipMemcpyAsync(c->d_pp, &c->h_pp, sizeof pp, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->h_pp_valid = true;
  }
  c->stats.index_kind = c->idx_kind == 1 ? (c->idx_wide ? 2u : 1u) : (c->idx_lines ? 3u : 0u);
  c->stats.match_variant = 0;  // (match_ctx_pass says which fused kernel it launches)
  c->stats.index_bytes = c->idx_kind == 1
                             ? ((1ull << c->idx_bits) + 1) * sizeof(CtxBucket) + ctx_entries_bytes(c->idx_novf + 16, c->idx_wide)
                             : ((1ull << c->idx_bits) + 1) * (c->idx_lines ? sizeof(LineBucket) : sizeof(Bucket)) + (c->idx_novf + 16) * sizeof(uint4);
  if (c->idx_kind == 1)
    return match_ctx_pass(c, P, pp, block_mode, block_thr, max_matches, planned_batches, nhits);

  Timer tm(c);
  hipEvent_t ev0 = pool_event(c), ev1 = pool_event(c);
  if (!ev0 || !ev1) return fail(c, 10, "hipEventCreate failed");
  HIPCHK(c, hipEventRecord(ev0, c->stream));

  const bool mask = c->reads_have_x || c->db_has_x;  // (a stale all-zero plane of an earlier batch does not count)
  // a mask plane on only one side: allocate the missing all-zero plane once
  if (mask && !c->rdm && c->nreads) {
    const uint64_t words = c->nreads * (uint64_t)c->rw;
    HIPCHK(c, hipMalloc((void**)&c->rdm, words * 4 + 256));
    HIPCHK(c, hipMemsetAsync(c->rdm, 0, words * 4 + 256, c->stream));
  }
  if (mask && !c->dbm2) {
    HIPCHK(c, hipMalloc((void**)&c->dbm2, (c->db_words + 64) * 4));
    // (the index stays valid: bucket_of treats a null and an all-zero mask plane alike)
    HIPCHK(c, hipMemsetAsync(c->dbm2, 0, (c->db_words + 64) * 4, c->stream));
    if ((rc = db_xblocks(c))) return rc;
  }

  // Per batch: k_screen claims descriptor space as it goes; if a batch needs more than the
  // buffer holds it reports how much and is repeated after growing the buffer (the first pass
  // of a workload sizes it).  A pass over the same reads, database and parameters as the last
  // completed one ("sized") is known to fit and runs without any host round trip; every kernel
  // still guards its writes, and the flags are checked once at the end.
  const bool sized = c->sized_epoch == c->data_epoch && c->sized_exact_blocks == (block_mode == 2) &&
                     memcmp(&c->sized_params, P, sizeof *P) == 0 && !c->env.debug_sync;
  uint64_t n_cand = 0, n_pairs = 0, n_windows = 0, n_two = 0;
  const uint64_t PAIR_CAP = 1ull << 31;  // u32 descriptor offsets
  uint64_t r0 = 0;
  uint32_t bsz = sized ? c->sized_bsz : c->batch_reads;
  if (sized) HIPCHK(c, hipMemsetAsync(c->counters + 8, 0, 8 * sizeof(unsigned long long), c->stream));
  // MUSC_PIPELINE=1: a sized pass of several batches is pipelined over two streams, k_screen of
  // batch b+1 beside k_confirm + k_compact of batch b, alternating between the two batch sets.
  // Off by default: measured on cfg3 / cfg4 / cfg5 shards the pass moves ~5.6 TB/s of cache lines
  // through HBM either way (both kernels are bound by the lines they fetch), so overlapping them
  // gains nothing (5.66 vs 5.44 ms on cfg3) and the second set costs memory.
  const bool want_pipe = c->env.pipeline;
  const bool piped = sized && want_pipe && c->nreads > bsz && c->bs[1].cdesc.cap >= c->bs[0].cdesc.cap;
  hipStream_t sA = c->stream, sB = piped ? c->stream2 : c->stream;
  c->s_confirm = sB;
  c->cur = 0;
  if (piped) {  // the confirm stream starts after the memsets above
    HIPCHK(c, hipEventRecord(c->ev_join, sA));
    HIPCHK(c, hipStreamWaitEvent(sB, c->ev_join, 0));
  }
  uint32_t batch_no = 0;
  while (r0 < c->nreads) {
    const uint32_t n = (uint32_t)std::min<uint64_t>(bsz, c->nreads - r0);
    if (piped) {
      c->cur = (int)(batch_no & 1u);
      // the set is free once the batch before last has been compacted
      if (batch_no >= 2) HIPCHK(c, hipStreamWaitEvent(sA, c->ev_free[c->cur], 0));
    }
    const int W = pp.W;
    const uint32_t ntiles = nblk(n, TILE);
    uint64_t total = 1;  // pairs of this batch (unknown on a sized pass)
    if (!sized) {
      if ((rc = ensure(c, c->bs[c->cur].wb, (uint64_t)n * W))) return rc;
      if ((rc = ensure(c, c->bs[c->cur].rvalid, (uint64_t)n + 1))) return rc;
      if ((rc = ensure(c, c->bs[c->cur].tbase, (uint64_t)ntiles + 1))) return rc;
      if ((rc = ensure(c, c->bs[c->cur].tcount, (uint64_t)ntiles + 1))) return rc;
      if ((rc = ensure(c, c->scan_tmp, scan_tmp_elems((uint64_t)ntiles + 1)))) return rc;
      if ((rc = ensure(c, c->tcount2, (uint64_t)ntiles + 1))) return rc;
      if ((rc = ensure(c, c->tpre, (uint64_t)ntiles + 1))) return rc;
      if ((rc = ensure(c, c->bs[c->cur].cdesc, std::max<uint64_t>(4ull * n, 1024)))) return rc;
      // batch-local counters: [0] valid windows [3] candidates [4] pairs [7] descriptor cursor
      HIPCHK(c, hipMemsetAsync(c->counters + 8, 0, 8 * sizeof(unsigned long long), c->stream));
    }

    if ((rc = upload_prepare(c, r0, n, c->stream))) return rc;  // (reads still on their way from the host)
    tm.begin(0);
    {
      Range rg("k_screen");
      launch_stage(c, 0, mask, r0, n, pp);
    }
    HIPCHK(c, hipGetLastError());
    tm.end(0);
    if (piped) {
      HIPCHK(c, hipEventRecord(c->ev_ready[c->cur], sA));
      HIPCHK(c, hipStreamWaitEvent(sB, c->ev_ready[c->cur], 0));
    }
    if (!sized) {
      HIPCHK(c, hipMemcpyAsync(&c->h_pinned[0], c->counters + 8, 8 * 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(&c->h_pinned[8], c->counters + 2, 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      total = c->h_pinned[4];
      const uint64_t hits_so_far = c->h_pinned[8];
      const uint64_t sgrid = screen_grid(c, n, mask);
      const uint64_t need = c->h_pinned[7] * sgrid;  // every workgroup region as large as the fullest
      if (need > PAIR_CAP) {
        // too many pairs for one launch: retry this range with half the reads
        if (n == 1) return fail(c, 6, "one read has %llu candidate pairs (> 2^31)", (unsigned long long)total);
        bsz = n / 2;
        HIPCHK(c, hipMemsetAsync(c->counters + 3, 0, 8, c->stream));
        continue;
      }
      if (c->h_pinned[7] > c->bs[c->cur].cdesc.cap / sgrid) {
        if ((rc = ensure(c, c->bs[c->cur].cdesc, need + need / 8 + sgrid))) return rc;
        HIPCHK(c, hipMemsetAsync(c->counters + 3, 0, 8, c->stream));
        continue;  // repeat the batch with room for every workgroup's pairs
      }
      n_windows += c->h_pinned[0];
      n_cand += c->h_pinned[3];
      n_pairs += c->h_pinned[4];
      n_two += c->h_pinned[5];
      if ((rc = ensure(c, c->p_nx, c->bs[c->cur].cdesc.cap))) return rc;
      if ((rc = ensure(c, c->stage, c->bs[c->cur].cdesc.cap))) return rc;
      if ((rc = ensure(c, c->hits, hits_so_far + total, true))) return rc;
    }
    c->stats.n_batches++;

    if (total) {
      const dim3 sg(std::min(nblk(n, TILE), MAX_GRID));

      tm.begin(3, sB);
      {
        Range rg("k_confirm");
        launch_stage(c, 2, mask, r0, n, pp);
      }
      HIPCHK(c, hipGetLastError());
      tm.end(3, sB);
      c->stats.confirm_launches++;

      Range rgc("scan + k_compact");
      tm.begin(4, sB);
      tm.begin(1, sB);
      rc = scan_u32(c, c->tcount2.p, c->tpre.p, (uint64_t)ntiles + 1, false, c->scan_tmp.p, sB);
      if (rc) return rc;
      tm.end(1, sB);
      hipLaunchKernelGGL(k_compact, sg, dim3(256), 0, sB, ntiles, c->bs[c->cur].tbase.p, c->tcount2.p, c->tpre.p,
                         c->stage.p, reinterpret_cast<uint4*>(c->hits.p), c->hits.cap, c->counters);
      HIPCHK(c, hipGetLastError());
      hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, sB, c->tpre.p, ntiles, c->counters);
      HIPCHK(c, hipGetLastError());
      tm.end(4, sB);
    }
    if (piped) HIPCHK(c, hipEventRecord(c->ev_free[c->cur], sB));
    r0 += n;
    batch_no++;
  }
  if (piped) {  // join: everything below is ordered after both streams
    HIPCHK(c, hipEventRecord(c->ev_join, sB));
    HIPCHK(c, hipStreamWaitEvent(sA, c->ev_join, 0));
  }
  c->cur = 0;
  c->s_confirm = c->stream;
  c->last_pp = pp;
  c->last_max_matches = (uint32_t)max_matches;
  c->last_exact_blocks = block_mode == 2;
  if (block_mode == 2) {
    hipLaunchKernelGGL(k_block_overflow, dim3(1024), dim3(256), 0, c->stream, c->block_table.p,
                       (uint32_t)max_matches, c->counters);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipEventRecord(ev1, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->counters, 16 * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (sized) {
    if (c->h_pinned[3]) {
      // a guard fired: the pass did not fit after all -- forget the sizing and run it the careful way
      c->sized_epoch = 0;
      return match_device_impl(c, P, nhits);
    }
    n_windows = c->h_pinned[8];  // the batch-local block accumulated over the whole pass
    n_cand = c->h_pinned[8 + 3];
    n_pairs = c->h_pinned[8 + 4];
    n_two = c->h_pinned[8 + 5];
  } else if (c->h_pinned[3]) {
    return fail(c, 12, "internal: a capacity guard fired although every batch was sized (flags %llu)",
                (unsigned long long)c->h_pinned[3]);
  }
  c->stats.n_read_windows = n_windows;
  c->stats.n_accepted = c->h_pinned[1];
  c->stats.n_hits = c->nhits = c->h_pinned[2];
  c->stats.n_descriptors = n_pairs;
  c->stats.n_pairs = n_pairs + n_two;  // a two-window descriptor is two of the reference's candidate pairs
  c->stats.n_candidates = n_cand;
  // 0 = proven: no (window,key) block exceeded MaxMatches, the tuples equal the reference's;
  // otherwise an upper bound on the number of such blocks (or ~0ull when the check was skipped)
  c->stats.n_overflow_blocks = check_blocks ? c->h_pinned[5] : ~0ull;
  if (block_mode == 1 && (c->h_pinned[6] || c->stats.n_batches > planned_batches)) {
    // screening inconclusive (a hot sketch cell, or more launches than the threshold assumed):
    // repeat the pass with exact per-block counters
    c->force_exact_blocks = true;
    c->exact_epoch = c->data_epoch;
    c->exact_params = *P;
    rc = match_device_impl(c, P, nhits);
    c->force_exact_blocks = false;
    return rc;
  }
  c->stats.ms_screen = tm.total(0);
  c->stats.ms_scan = tm.total(1);
  c->stats.ms_confirm = tm.total(3);
  c->stats.ms_select = tm.total(4);
  (void)hipEventElapsedTime(&c->stats.ms_total, ev0, ev1);
  // 12 B descriptor + ceil(2L/8) B read + ceil(2L/8)+1 B target span per pair the
  // launch loads, + 16 B per tuple written.  A launch loads one descriptor, one record and one
  // span per DESCRIPTOR; a descriptor that stands for two windows is two of the reference's
  // candidate pairs but is fetched once, so the bytes are billed per descriptor (the r01 figure
  // billed them per pair and over-credited the kernel).
  const uint64_t L = c->max_len;
  c->stats.confirm_bytes = c->stats.n_descriptors * (12 + (2 * L + 7) / 8 + (2 * L + 7) / 8 + 1) + 16 * c->stats.n_hits;
  if (nhits) *nhits = c->nhits;
  if (!sized && want_pipe) {  // the second batch set gets the capacities the first one ended up with
    if ((rc = ensure(c, c->bs[1].wb, c->bs[0].wb.cap)) || (rc = ensure(c, c->bs[1].rvalid, c->bs[0].rvalid.cap)) ||
        (rc = ensure(c, c->bs[1].tbase, c->bs[0].tbase.cap)) || (rc = ensure(c, c->bs[1].tcount, c->bs[0].tcount.cap)) ||
        (rc = ensure(c, c->bs[1].cdesc, c->bs[0].cdesc.cap)))
      return rc;
  }
  c->sized_epoch = c->data_epoch;
  c->sized_params = *P;
  c->sized_exact_blocks = block_mode == 2;
  c->sized_bsz = bsz;
  return 0;
}

int musc_hits_copy(musc_ctx* c, musc_hit* dst, uint64_t capacity, int dst_on_device) {
  if (!c) return 1;
  if (capacity < c->nhits) return fail(c, 2, "musc_hits_copy: capacity %llu < %llu hits",
                                       (unsigned long long)capacity, (unsigned long long)c->nhits);
  if (c->nhits == 0) return 0;
  if (!dst) return fail(c, 2, "musc_hits_copy: dst is NULL");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(dst, c->hits.p, c->nhits * sizeof(musc_hit),
                           dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

static int check_pack_bits(musc_ctx* c, const int32_t* bits, PackBits* b) {
  if (!bits) return fail(c, 2, "bits is NULL");
  int sum = 0;
  for (int i = 0; i < 4; i++) {
    if (bits[i] < 1 || bits[i] > 32) return fail(c, 2, "field width %d outside 1..32", bits[i]);
    sum += bits[i];
  }
  if (sum > 64) return fail(c, 2, "field widths add up to %d > 64 bits", sum);
  *b = PackBits{bits[0], bits[1], bits[2], bits[3]};
  return 0;
}

int musc_hits_copy_packed(musc_ctx* c, uint64_t* dst, uint64_t capacity, int dst_on_device, uint64_t read_base,
                          const int32_t* bits) {
  if (!c) return 1;
  PackBits b;
  int rc = check_pack_bits(c, bits, &b);
  if (rc) return rc;
  if (capacity < c->nhits) return fail(c, 2, "musc_hits_copy_packed: capacity %llu < %llu hits",
                                       (unsigned long long)capacity, (unsigned long long)c->nhits);
  if (c->nhits == 0) return 0;
  if (!dst) return fail(c, 2, "musc_hits_copy_packed: dst is NULL");
  HIPCHK(c, hipSetDevice(c->device));
  uint64_t* out = dst;
  if (!dst_on_device) {
    if ((rc = ensure(c, c->packed, c->nhits))) return rc;
    out = c->packed.p;
  }
  HIPCHK(c, hipMemsetAsync(c->d_flag, 0, 4, c->stream));
  hipLaunchKernelGGL(k_pack_hits, dim3(std::min(nblk(c->nhits, 256), MAX_GRID)), dim3(256), 0, c->stream,
                     reinterpret_cast<const uint4*>(c->hits.p), c->nhits, read_base, b, out, c->d_flag);
  HIPCHK(c, hipGetLastError());
  uint32_t bad = 0;
  HIPCHK(c, hipMemcpyAsync(&bad, c->d_flag, 4, hipMemcpyDeviceToHost, c->stream));
  if (!dst_on_device)
    HIPCHK(c, hipMemcpyAsync(dst, out, c->nhits * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (bad) return fail(c, 8, "musc_hits_copy_packed: a tuple field does not fit its width (%d/%d/%d/%d bits)",
                       b.read, b.gene, b.pos, b.nmiss);
  return 0;
}

int musc_hits_copy_compact(musc_ctx* c, uint32_t* words, uint64_t words_cap, uint8_t* counts, uint64_t counts_cap,
                           int dst_on_device, const int32_t* bits) {
  if (!c) return 1;
  if (!bits) return fail(c, 2, "bits is NULL");
  int sum = 0;
  for (int i = 0; i < 3; i++) {
    if (bits[i] < 1 || bits[i] > 30) return fail(c, 2, "field width %d outside 1..30", bits[i]);
    sum += bits[i];
  }
  if (sum > 32) return fail(c, 2, "field widths add up to %d > 32 bits", sum);
  if (words_cap < c->nhits) return fail(c, 2, "musc_hits_copy_compact: room for %llu tuples < %llu",
                                        (unsigned long long)words_cap, (unsigned long long)c->nhits);
  if (counts_cap < c->nreads) return fail(c, 2, "musc_hits_copy_compact: room for %llu reads < %llu",
                                          (unsigned long long)counts_cap, (unsigned long long)c->nreads);
  if ((c->nhits && !words) || (c->nreads && !counts)) return fail(c, 2, "musc_hits_copy_compact: NULL destination");
  HIPCHK(c, hipSetDevice(c->device));
  const PackBits b{0, bits[0], bits[1], bits[2]};
  uint32_t* dw = words;
  uint8_t* dc = counts;
  if (!dst_on_device) {  // stage on the device, then one copy each
    int rc = ensure(c, c->packed, (c->nhits * 4 + c->nreads + 15) / 8 + 2);
    if (rc) return rc;
    dw = reinterpret_cast<uint32_t*>(c->packed.p);
    dc = reinterpret_cast<uint8_t*>(dw + c->nhits);
  }
  HIPCHK(c, hipMemsetAsync(c->d_flag, 0, 4, c->stream));
  if (c->nreads) HIPCHK(c, hipMemsetAsync(dc, 0, c->nreads, c->stream));
  if (c->nhits) {
    hipLaunchKernelGGL(k_pack_compact, dim3(std::min(nblk(c->nhits, 256), MAX_GRID)), dim3(256), 0, c->stream,
                       reinterpret_cast<const uint4*>(c->hits.p), c->nhits, b, dw, dc, c->d_flag);
    HIPCHK(c, hipGetLastError());
  }
  uint32_t bad = 0;
  HIPCHK(c, hipMemcpyAsync(&bad, c->d_flag, 4, hipMemcpyDeviceToHost, c->stream));
  if (!dst_on_device) {
    if (c->nhits) HIPCHK(c, hipMemcpyAsync(words, dw, c->nhits * 4, hipMemcpyDeviceToHost, c->stream));
    if (c->nreads) HIPCHK(c, hipMemcpyAsync(counts, dc, c->nreads, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (bad & 4u) return fail(c, 12, "internal: the hit list is not read-major");
  if (bad) return fail(c, 8, "musc_hits_copy_compact: %s", (bad & 1u) ? "a tuple field does not fit its width"
                                                                      : "a read has more than 255 tuples");
  return 0;
}

int musc_hits_unpack(musc_ctx* c, const uint64_t* src, uint64_t n, int on_device, const int32_t* bits, musc_hit* dst) {
  if (!c) return 1;
  PackBits b;
  int rc = check_pack_bits(c, bits, &b);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!src || !dst) return fail(c, 2, "musc_hits_unpack: NULL pointer");
  if (!on_device) {  // host to host: plain loop, same layout
    for (uint64_t i = 0; i < n; i++) {
      uint64_t v = src[i];
      dst[i].nmiss = (uint32_t)(v & ((1ull << b.nmiss) - 1ull));
      v >>= b.nmiss;
      dst[i].pos = (uint32_t)(v & ((1ull << b.pos) - 1ull));
      v >>= b.pos;
      dst[i].gene_idx = (uint32_t)(v & ((1ull << b.gene) - 1ull));
      v >>= b.gene;
      dst[i].read_idx = (uint32_t)v;
    }
    return 0;
  }
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(k_unpack_hits, dim3(std::min(nblk(n, 256), MAX_GRID)), dim3(256), 0, c->stream, src, n, b,
                     reinterpret_cast<uint4*>(dst));
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int musc_match(musc_ctx* c, const musc_params* P, musc_hit** hits, uint64_t* nhits) {
  if (!c) return 1;
  if (!hits || !nhits) return fail(c, 2, "musc_match: NULL output pointer");
  *hits = nullptr;
  *nhits = 0;
  uint64_t n = 0;
  int rc = musc_match_device(c, P, &n);
  if (rc) return rc;
  musc_hit* h = (musc_hit*)malloc(sizeof(musc_hit) * (n ? n : 1));
  if (!h) return fail(c, 7, "out of host memory for %llu hits", (unsigned long long)n);
  rc = musc_hits_copy(c, h, n, 0);
  if (rc) {
    free(h);
    return rc;
  }
  *hits = h;
  *nhits = n;
  return 0;
}

void musc_free_hits(musc_hit* hits) { free(hits); }

int musc_overflow_probes(musc_ctx* c, uint32_t** read_idx, uint32_t** window, uint64_t* n) {
  if (!c) return 1;
  if (!read_idx || !window || !n) return fail(c, 2, "musc_overflow_probes: NULL output pointer");
  *read_idx = *window = nullptr;
  *n = 0;
  if (c->stats.n_overflow_blocks == 0 || c->stats.n_overflow_blocks == ~0ull) return 0;
  if (!c->last_exact_blocks || !c->block_table.p) return fail(c, 4, "no exact block counters from the last pass");
  HIPCHK(c, hipSetDevice(c->device));
  uint64_t cap = 1u << 20;
  for (;;) {
    TmpBufs B;  // released at the end of every iteration and on every return
    uint2* d_out = nullptr;
    HIPCHK(c, B.alloc(&d_out, cap * sizeof(uint2)));
    HIPCHK(c, hipMemsetAsync(c->counters + 8, 0, 8, c->stream));
    const dim3 grid(std::min(nblk(c->nreads, 256), MAX_GRID)), block(256);
    switch (c->rw) {
      case 4: hipLaunchKernelGGL((k_hot_probes<4>), grid, block, 0, c->stream, c->rd, c->rdm, c->nreads, c->rw, c->d_pp, c->block_table.p, c->last_max_matches, d_out, cap, c->counters + 8); break;
      case 8: hipLaunchKernelGGL((k_hot_probes<8>), grid, block, 0, c->stream, c->rd, c->rdm, c->nreads, c->rw, c->d_pp, c->block_table.p, c->last_max_matches, d_out, cap, c->counters + 8); break;
      case 12: hipLaunchKernelGGL((k_hot_probes<12>), grid, block, 0, c->stream, c->rd, c->rdm, c->nreads, c->rw, c->d_pp, c->block_table.p, c->last_max_matches, d_out, cap, c->counters + 8); break;
      case 16: hipLaunchKernelGGL((k_hot_probes<16>), grid, block, 0, c->stream, c->rd, c->rdm, c->nreads, c->rw, c->d_pp, c->block_table.p, c->last_max_matches, d_out, cap, c->counters + 8); break;
      default: hipLaunchKernelGGL((k_hot_probes<0>), grid, block, 0, c->stream, c->rd, c->rdm, c->nreads, c->rw, c->d_pp, c->block_table.p, c->last_max_matches, d_out, cap, c->counters + 8); break;
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_pinned, c->counters + 8, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, 10, "musc_overflow_probes: %s", hipGetErrorString(e));
    const uint64_t found = c->h_pinned[0];
    if (found > cap) {  // retry with room for all of them
      cap = found + 16;
      continue;
    }
    std::vector<uint2> h(found ? found : 1);
    if (found) e = hipMemcpy(h.data(), d_out, found * sizeof(uint2), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(c, 10, "musc_overflow_probes: %s", hipGetErrorString(e));
    uint32_t* r = (uint32_t*)malloc(sizeof(uint32_t) * (found ? found : 1));
    uint32_t* w = (uint32_t*)malloc(sizeof(uint32_t) * (found ? found : 1));
    if (!r || !w) {
      free(r);
      free(w);
      return fail(c, 7, "musc_overflow_probes: out of host memory");
    }
    for (uint64_t j = 0; j < found; j++) {
      r[j] = h[j].x;
      w[j] = h[j].y;
    }
    *read_idx = r;
    *window = w;
    *n = found;
    return 0;
  }
}

void musc_free_u32(uint32_t* p) { free(p); }

int musc_get_stats(musc_ctx* c, musc_stats* out) {
  if (!c || !out) return 1;
  *out = c->stats;
  return 0;
}

int musc_gather(musc_ctx* const* ctxs, int n, const uint64_t* read_base, musc_hit** hits, uint64_t* nhits) {
  if (!ctxs || n < 1 || !hits || !nhits) return 1;
  uint64_t total = 0;
  for (int i = 0; i < n; i++) {
    if (!ctxs[i]) return 1;
    total += ctxs[i]->nhits;
  }
  musc_hit* h = (musc_hit*)malloc(sizeof(musc_hit) * (total ? total : 1));
  if (!h) return fail(ctxs[0], 7, "musc_gather: out of host memory");
  uint64_t o = 0;
  for (int i = 0; i < n; i++) {
    musc_ctx* c = ctxs[i];
    int rc = musc_hits_copy(c, h + o, c->nhits, 0);
    if (rc) {
      free(h);
      return rc;
    }
    const uint64_t base = read_base ? read_base[i] : 0;
    for (uint64_t j = 0; j < c->nhits; j++) h[o + j].read_idx += (uint32_t)base;
    o += c->nhits;
  }
  *hits = h;
  *nhits = total;
  return 0;
}

// ---- RCCL, resolved at run time: a single-GPU user never loads the library, and a box without
// it still runs everything but musc_gather_rccl
namespace {
struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;
  std::map<std::vector<int>, std::vector<ncclComm_t>> comms;  // one clique per device list, kept for the process
};
std::mutex g_rccl_mu;

RcclApi* rccl_api() {
  static RcclApi api;
  if (api.lib || !api.err.empty()) return &api;
  // MUSC_RCCL_LIB names the library (tests point it at a missing file); otherwise the one next to the
  // HIP runtime this library links against, then whatever the loader finds
  std::string first_err;
  std::vector<std::string> names;
  if (const char* e = getenv("MUSC_RCCL_LIB")) names.push_back(e);
  else names = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
  for (const std::string& name : names) {
    (void)dlerror();
    api.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (api.lib) break;
    const char* e = dlerror();  // (one call: it returns the message and clears it)
    if (first_err.empty()) first_err = e ? e : "not found";
  }
  if (!api.lib) {
    api.err = std::string("cannot load librccl: ") + first_err;
    return &api;
  }
#define MUSC_RCCL_SYM(FIELD, NAME)                                   \
  api.FIELD = reinterpret_cast<decltype(api.FIELD)>(dlsym(api.lib, NAME)); \
  if (!api.FIELD && api.err.empty()) api.err = std::string("librccl lacks ") + NAME;
  MUSC_RCCL_SYM(CommInitAll, "ncclCommInitAll")
  MUSC_RCCL_SYM(CommDestroy, "ncclCommDestroy")
  MUSC_RCCL_SYM(GroupStart, "ncclGroupStart")
  MUSC_RCCL_SYM(GroupEnd, "ncclGroupEnd")
  MUSC_RCCL_SYM(Send, "ncclSend")
  MUSC_RCCL_SYM(Recv, "ncclRecv")
  MUSC_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef MUSC_RCCL_SYM
  return &api;
}
}  // namespace

int musc_rccl_probe(char* msg, uint64_t cap) {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  RcclApi* api = rccl_api();
  if (msg && cap) {
    strncpy(msg, api->err.c_str(), cap - 1);
    msg[cap - 1] = 0;
  }
  return api->err.empty() ? 0 : 20;
}

__global__ void k_rebase_reads(uint4* __restrict__ h, uint64_t n, uint32_t base) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) h[i].x += base;
}

int musc_gather_rccl(musc_ctx* const* ctxs, int n, const uint64_t* read_base, musc_hit** hits, uint64_t* nhits) {
  if (!ctxs || n < 1 || !hits || !nhits) return 1;
  *hits = nullptr;
  *nhits = 0;
  uint64_t total = 0;
  std::vector<uint64_t> off(n + 1, 0);
  std::vector<int> devs(n);
  for (int i = 0; i < n; i++) {
    if (!ctxs[i]) return 1;
    devs[i] = ctxs[i]->device;
    off[i] = total;
    total += ctxs[i]->nhits;
  }
  off[n] = total;
  musc_ctx* c0 = ctxs[0];
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++)
      if (devs[i] == devs[j]) return fail(c0, 2, "musc_gather_rccl: contexts %d and %d share device %d", i, j, devs[i]);
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  std::vector<ncclComm_t>* comms = nullptr;
  RcclApi* api = nullptr;
  if (n > 1) {
    api = rccl_api();
    if (!api->err.empty()) return fail(c0, 20, "musc_gather_rccl: %s", api->err.c_str());
    auto it = api->comms.find(devs);
    if (it == api->comms.end()) {
      std::vector<ncclComm_t> cs(n);
      const ncclResult_t r = api->CommInitAll(cs.data(), n, devs.data());
      if (r != ncclSuccess) return fail(c0, 20, "musc_gather_rccl: ncclCommInitAll: %s", api->GetErrorString(r));
      it = api->comms.emplace(devs, cs).first;
    }
    comms = &it->second;
  }
  HIPCHK(c0, hipSetDevice(c0->device));
  int rc = ensure(c0, c0->gathered, total ? total : 1);
  if (rc) return rc;
  // the destination's own tuples never touch a link; everything else arrives over xGMI, all
  // transfers in ONE group so that the links run side by side
  if (c0->nhits)
    HIPCHK(c0, hipMemcpyAsync(c0->gathered.p, c0->hits.p, c0->nhits * sizeof(musc_hit), hipMemcpyDeviceToDevice, c0->stream));
  if (n > 1) {
    ncclResult_t r = api->GroupStart();
    for (int i = 1; i < n && r == ncclSuccess; i++) {
      if (!ctxs[i]->nhits) continue;
      const size_t bytes = ctxs[i]->nhits * sizeof(musc_hit);
      (void)hipSetDevice(ctxs[i]->device);
      r = api->Send(ctxs[i]->hits.p, bytes, ncclUint8, 0, (*comms)[i], ctxs[i]->stream);
      if (r != ncclSuccess) break;
      (void)hipSetDevice(c0->device);
      r = api->Recv(c0->gathered.p + off[i], bytes, ncclUint8, i, (*comms)[0], c0->stream);
    }
    const ncclResult_t r2 = api->GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return fail(c0, 20, "musc_gather_rccl: %s", api->GetErrorString(r));
  }
  HIPCHK(c0, hipSetDevice(c0->device));
  for (int i = 0; i < n; i++) {
    const uint64_t base = read_base ? read_base[i] : 0;
    if (!base || !ctxs[i]->nhits) continue;
    hipLaunchKernelGGL(k_rebase_reads, dim3(std::min(nblk(ctxs[i]->nhits, 256), MAX_GRID)), dim3(256), 0, c0->stream,
                       reinterpret_cast<uint4*>(c0->gathered.p + off[i]), ctxs[i]->nhits, (uint32_t)base);
    HIPCHK(c0, hipGetLastError());
  }
  for (int i = 1; i < n; i++) {  // the senders' streams
    HIPCHK(ctxs[i], hipSetDevice(ctxs[i]->device));
    HIPCHK(ctxs[i], hipStreamSynchronize(ctxs[i]->stream));
  }
  HIPCHK(c0, hipSetDevice(c0->device));
  musc_hit* h = (musc_hit*)malloc(sizeof(musc_hit) * (total ? total : 1));
  if (!h) return fail(c0, 7, "musc_gather_rccl: out of host memory");
  hipError_t e = hipSuccess;
  if (total) e = hipMemcpyAsync(h, c0->gathered.p, total * sizeof(musc_hit), hipMemcpyDeviceToHost, c0->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c0->stream);
  if (e != hipSuccess) {
    free(h);
    return fail(c0, 10, "musc_gather_rccl: %s", hipGetErrorString(e));
  }
  *hits = h;
  *nhits = total;
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------- read prep (sort + collapse)
#include "muscato_prep.hpp"
