/* muscato_hip.h -- C ABI of libmuscato_hip.so: muscato's seed-and-extend hot path
 * (muscato_screen -> muscato_confirm [-> per-read best+MMTol]) on AMD MI355X (gfx950).
 *
 * The reference (kshedden/muscato, pure Go) has no in-process plugin/FFI interface for
 * this path: its boundary is two executables joined by snappy text files in TempDir,
 *     exec.Command("muscato_screen",  config.json)        cmd/muscato/main.go:306-316
 *     GNU sort of bmatch_k -> smatch_k                    cmd/muscato/main.go:318-385
 *     exec.Command("muscato_confirm", config.json, k)     cmd/muscato/main.go:387-420
 * with inputs  reads_sorted.txt.sz (unique reads), GeneFileName (one target per line,
 * gene number = line index) and outputs rmatch_k.txt.sz, lines
 * "read \t targetsub \t pos \t nmiss \t %011d"   (cmd/muscato_confirm/main.go:221-230).
 * This header is the boundary a cgo (or any FFI) host binds instead of spawning those
 * processes; INTEGRATION.md shows the cgo stub.  Each entry point names the reference
 * code it replaces.
 *
 * Conventions
 *   - plain C types only; no HIP / torch types cross the boundary.
 *   - every function returning int returns 0 on success, non-zero on error; the text of
 *     the last error of a context is musc_last_error(ctx).  Nothing aborts or throws
 *     across the boundary (the reference convention is exit status != 0,
 *     cmd/muscato/main.go:313-315, 411-415 -- the CLI maps errors to that).
 *   - ownership: inputs are borrowed for the duration of the call; outputs returned
 *     through musc_hit** are owned by the library until musc_free_hits().
 *   - threading: one musc_ctx per GPU; calls on one ctx must be serialised by the caller;
 *     different ctxs may be driven from different host threads (a Go caller must
 *     runtime.LockOSThread() around a call sequence: HIP's current device is per thread).
 *   - there is NO CPU fallback: if no gfx950 device is usable musc_init fails.
 */
#ifndef MUSCATO_HIP_H
#define MUSCATO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MUSC_ABI_VERSION 3
#define MUSC_MAX_WINDOWS 16

typedef struct musc_ctx musc_ctx;

/* One accepted placement: read `read_idx` (index into the loaded unique reads) matches
 * target `gene_idx` (line index of the gene file) at target offset `pos` with `nmiss`
 * mismatches.  Equivalent to one rmatch_k / matches.txt line
 * (cmd/muscato_confirm/main.go:221-230). */
typedef struct {
  uint32_t read_idx;
  uint32_t gene_idx;
  uint32_t pos;
  uint32_t nmiss;
} musc_hit;

/* The part of utils.Config (utils/config.go:10-101) the hot path reads. */
typedef struct {
  int32_t n_windows;                 /* len(Config.Windows), 1..MUSC_MAX_WINDOWS       */
  int32_t windows[MUSC_MAX_WINDOWS]; /* Config.Windows: start of each window in a read */
  int32_t window_width;              /* Config.WindowWidth                             */
  double pmatch;                     /* Config.PMatch; nmiss = int((1-PMatch)*len)     */
  int32_t min_dinuc;                 /* Config.MinDinuc (utils/entropy.go:5-40 gate)   */
  int32_t max_read_length;           /* Config.MaxReadLength                           */
  int32_t max_matches;               /* Config.MaxMatches (per window-key block)       */
  int32_t match_mode;                /* 0 = "best", 1 = "first" (Config.MatchMode)     */
  int32_t mmtol;                     /* Config.MMTol                                   */
  /* 0: return every accepted tuple = set-union of rmatch_0..W-1 (output of muscato_confirm
   *    after combine_filter | sort -u, cmd/muscato/main.go:441-463);
   * 1: additionally keep, per read, only nmiss <= best+MMTol = matches.txt
   *    (cmd/muscato_combine_windows/main.go:36-60). */
  int32_t apply_mmtol;
  /* Addition (the reference has no such flag; BASELINE.json's --MaxMismatch): 0 = budget from
   * PMatch as above; v > 0 = every read may have at most v-1 mismatches. */
  int32_t max_mismatch_p1;
  /* 1 = skip the per-block MaxMatches check (musc_stats.n_overflow_blocks becomes ~0) */
  int32_t skip_block_check;
  /* > 1 when this context only sees 1/n_shards of the reads of a (window,key) block: the
   * MaxMatches check then uses MaxMatches / n_shards, so that "no overflow on any shard"
   * proves "no overflow" (SURVEY.md 8e caveat) */
  int32_t n_shards;
  int32_t reserved[2];
} musc_params;

/* Counters and device timings of the last musc_match* call on a context. */
typedef struct {
  uint64_t n_reads;         /* unique reads processed                                  */
  uint64_t n_read_windows;  /* (read, window) seeds that passed the length+entropy gate */
  uint64_t n_candidates;    /* index entries probed (k-mer hits incl. chance hits)      */
  uint64_t n_pairs;         /* (window, read, placement) candidate pairs that reached the
                             * confirm kernel -- the reference's smatch/win join output,
                             * minus what the flank filter already ruled out               */
  uint64_t n_accepted;      /* pairs with nmiss <= budget, after the union over windows */
  uint64_t n_hits;          /* tuples returned                                          */
  /* (window,key) blocks whose accepted pairs may exceed MaxMatches: 0 = proven none (the
   * reference's truncation, cmd/muscato_confirm/main.go:233-242, 424-448, never triggered and
   * the tuples are exact); > 0 = upper bound; ~0 = check skipped */
  uint64_t n_overflow_blocks;
  uint64_t confirm_bytes;   /* algorithmic bytes of the confirm launches: 63 B per DESCRIPTOR
                             * loaded at 100 bp (descriptor 12 + record 25 + target span 26,
                             * SURVEY.md 8d) + 16 B per tuple written                       */
  uint32_t confirm_launches;
  uint32_t n_batches;
  float ms_screen;          /* HIP-event time of each kernel family, summed over batches:  */
  float ms_scan;            /*   k_screen (index_kind 0) or k_match (1) | scan | (see match_variant) |
                             *   k_confirm (index_kind 0 only) | scan+k_compact            */
  uint32_t match_variant;   /* (ABI 3; the slot was an unused float) which fused kernel ran on context buckets:
                             * 0 = none (two-kernel path), 2 = k_match_t, general
                             * instance, 3 = k_match_t specialised for the run's geometry (SpecGeom<1>),
                             * 4 = k_match_g (three waves per SIMD, LDS-DMA), general, 5 = k_match_g specialised */
  float ms_confirm;
  float ms_select;
  float ms_total;           /* first launch to last completion on the context's stream  */
  float ms_index_build;     /* last musc_db_build_index                                 */
  float ms_read_prep;       /* last musc_reads_sort_unique (device time)                   */
  uint64_t n_descriptors;   /* descriptors k_screen wrote: one per placement of a read, also
                             * when two windows of the read found it (then it is two pairs);
                             * 0 with context buckets (no descriptors exist)               */
  /* ---- since ABI version 2 */
  uint32_t index_kind;      /* index the pass ran on: 0 = 64-byte window-start buckets + target
                             * gather (k_screen -> k_confirm), 3 = the same on 128-byte line
                             * buckets (dense databases; k_screen_t -> k_confirm), 1 = context
                             * buckets, 120 bases (k_match_t or k_match), 2 = wide context
                             * buckets, 200 bases (k_match_t)                                */
  uint32_t match_launches;  /* launches of that kernel (index_kind 1)                       */
  uint64_t n_overflow_entries; /* index entries beyond a bucket's inline ones that were walked */
  uint64_t match_bytes;     /* algorithmic bytes of those launches: record (ceil(2L/8) B)
                             * per read + one 128-B bucket line per probe + 40 B (wide: 60)
                             * per overflow entry walked + 16 B per tuple staged                         */
  uint64_t match_bytes_strict; /* the same with a probe billed for what it uses of its line:
                             * 8 B header + 40 B per inline entry present                   */
  uint64_t index_bytes;     /* device memory held by the index (table + overflow entries)  */
} musc_stats;

int musc_abi_version(void);

/* Create / destroy a context bound to HIP device `device_ordinal` (one per GPU). */
int musc_init(int device_ordinal, musc_ctx** out);
void musc_destroy(musc_ctx* ctx);
const char* musc_last_error(musc_ctx* ctx); /* ctx may be NULL: error of a failed musc_init */
/* The MUSC_* environment knobs (tests, A/B runs: MUSC_INDEX, MUSC_MATCH, MUSC_GRAPH ...) are read once, by
 * musc_init; a pass never calls getenv.  This re-reads them for a live context (a test hook; the next pass sizes
 * its buffers again).  The reference has no counterpart: its knobs are the Config fields. */
int musc_reload_env(musc_ctx* ctx);

/* ---- target database: replaces muscato_screen's scan of GeneFileName
 * (cmd/muscato_screen/main.go:408-452; gene number = sequence index). --------------------
 *
 * ASCII form: `seqs` holds nseq sequences back to back, sequence i = bytes
 * [offsets[i], offsets[i+1]).  'A','C','G','T' are bases, every other byte is the
 * reference's 'X' (cmd/muscato_prep_targets/main.go:68-80).  on_device != 0 means both
 * pointers are device pointers. */
int musc_db_load_ascii(musc_ctx* ctx, const char* seqs, const uint64_t* offsets, uint32_t nseq,
                       int on_device);
/* Packed form: 2 bits per base (A=0 C=1 G=2 T=3), base j of the concatenated stream in bits
 * [2j%8, 2j%8+2) of byte j/4; nmask (may be NULL) has 1 bit per base, bit j%8 of byte j/8, set
 * where the base is 'X' (the 2-bit code under a set mask bit is ignored).  seq_offsets are in
 * bases, nseq+1 entries.  Host pointers. */
int musc_db_load_packed(musc_ctx* ctx, const uint8_t* bases2bit, const uint8_t* nmask,
                        const uint64_t* seq_offsets, uint32_t nseq);
/* Build the k-mer -> target position index for WindowWidth (done lazily by musc_match if the
 * width changed).  One-off per (database, width); replaces the per-run Bloom sketch + full
 * database scan of cmd/muscato_screen/main.go:116-207, 256-366 (result-equivalent: SURVEY.md
 * 8a note H). */
int musc_db_build_index(musc_ctx* ctx, int32_t window_width);
/* The same for a whole parameter block: builds the index musc_match* will pick for `params` and
 * reads of at most max_read_len bases (0 = params->max_read_length) -- context buckets (128-byte
 * buckets that carry each placement's 120 surrounding target bases, so that screen and confirm
 * are one kernel and no target gather is needed, or 200 bases for two placements per bucket; see
 * kernels_match.hpp) when the run fits them -- a database with X included: its entries whose
 * context holds an X are flagged and compared as cmd/muscato_confirm/main.go:151-159 does -- the
 * window-start-bucket index of musc_db_build_index otherwise.  musc_match* does this lazily;
 * calling it first only moves the one-off cost out of the first match.  MUSC_INDEX=classic in
 * the environment forces the window-start buckets, MUSC_NO_X_CONTEXT=1 only for runs with X. */
int musc_db_build_index_for(musc_ctx* ctx, const musc_params* params, int32_t max_read_len);

/* ---- reads: replaces reading reads_sorted.txt.sz (cmd/muscato_screen/main.go:120-191,
 * cmd/muscato_window_reads/main.go:94-141).  Reads must already be prepared as the
 * reference does (non-ACGT -> X, truncated to MaxReadLength, de-duplicated); read_idx in the
 * hits is the index into this array. --------------------------------------------------- */
int musc_reads_load_ascii(musc_ctx* ctx, const char* seqs, const uint64_t* offsets,
                          uint64_t nreads, int on_device);
int musc_reads_load_packed(musc_ctx* ctx, const uint8_t* bases2bit, const uint8_t* nmask,
                           const uint64_t* read_offsets, uint64_t nreads);
/* The same with 32-bit lengths instead of 64-bit offsets (`const uint32_t*
 * lengths_or_offsets`): lengths[i] = bases of read i, or lengths == NULL: every read has fixed_len
 * bases (nothing but the bases crosses PCIe then).  The reads sit back to back in the 2-bit stream.
 * async != 0 (fixed length, no mask): the call returns once the upload is QUEUED -- in pieces, on a
 * copy stream of its own -- and the next musc_match* on the context packs and matches each batch of
 * reads as its piece arrives, so that upload and matching overlap (the reference's equivalent is
 * muscato_screen reading reads_sorted.txt.sz while it hashes, cmd/muscato_screen/main.go:116-207).
 * The caller's buffer must then stay valid until that musc_match* call (or musc_destroy) returns;
 * pinned memory makes the upload asynchronous in fact. */
int musc_reads_load_packed32(musc_ctx* ctx, const uint8_t* bases2bit, const uint8_t* nmask,
                             const uint32_t* lengths, uint32_t fixed_len, uint64_t nreads, int async);

/* Read prep on the GPU: replaces the bytewise sort of the prepared reads (sortReads in
 * cmd/muscato/main.go: GNU sort of the `seq\tname` lines under LC_ALL=C) and the collapse of
 * identical sequences (cmd/muscato_uniqify/main.go:83-135) for the sequence column.
 * Input: nreads reads as they leave muscato_prep_reads (non-ACGT -> X, length filter, truncated),
 * in input order, ASCII + offsets like musc_reads_load_ascii.
 * Afterwards the context's reads are the DISTINCT sequences in bytewise order (a proper prefix
 * first), exactly as if the first column of reads_sorted.txt.sz had been loaded, and
 *   order[nreads]       input read numbers sorted by sequence, ties in input order
 *   ustart[*nunique+1]  distinct sequence g = the input reads order[ustart[g] .. ustart[g+1])
 * so the host can write `seq\tcount\tnames` (count = group size; the reference's name order within
 * a group is the bytewise order of the names, which the caller applies).  Both arrays are
 * malloc'ed by the library: musc_free_u32. */
int musc_reads_sort_unique(musc_ctx* ctx, const char* seqs, const uint64_t* offsets, uint64_t nreads,
                           int on_device, uint32_t** order, uint32_t** ustart, uint64_t* nunique);

/* ---- the hot path: screen + confirm (+ per-read best filter) -------------------------
 * musc_match_device leaves the hits in device memory (count in *nhits);
 * musc_hits_copy copies them to `dst` (host, or device if dst_on_device) -- capacity in
 * hits; musc_match = both, into a library-owned host array.  Hit order is unspecified. */
int musc_match_device(musc_ctx* ctx, const musc_params* params, uint64_t* nhits);
int musc_hits_copy(musc_ctx* ctx, musc_hit* dst, uint64_t capacity, int dst_on_device);
/* The same tuples as one u64 each, for the wire (gather to rank 0 over RCCL/xGMI) or for keeping
 * them compact: (read_idx + read_base) in the top bits, then gene_idx, pos, nmiss, with
 * bits[4] = the widths of read, gene, pos, nmiss (each 1..32, sum <= 64).  The numeric order of
 * the words is the lexicographic order of the tuples.  Fails with code 8 if a field of some
 * tuple does not fit its width.  musc_hits_unpack inverts it (src and dst both on the device or
 * both on the host). */
int musc_hits_copy_packed(musc_ctx* ctx, uint64_t* dst, uint64_t capacity, int dst_on_device,
                          uint64_t read_base, const int32_t* bits);
int musc_hits_unpack(musc_ctx* ctx, const uint64_t* src, uint64_t n, int on_device, const int32_t* bits,
                     musc_hit* dst);
/* The most compact wire form (5 bytes per tuple at one tuple per read, against 8 and 16): the hit
 * list is read-major -- a read's tuples are contiguous and reads come in increasing order -- so the
 * read index travels as counts[r] = number of tuples of read r (one byte per loaded read) and a
 * tuple is one u32 word gene | pos | nmiss with bits[3] = the widths of gene, pos, nmiss (sum <= 32).
 * Fails with code 8 if a field does not fit or a read has more than 255 tuples (use the 8-byte
 * form then).  words: room for nhits words, counts: room for the loaded reads. */
int musc_hits_copy_compact(musc_ctx* ctx, uint32_t* words, uint64_t words_capacity, uint8_t* counts,
                           uint64_t counts_capacity, int dst_on_device, const int32_t* bits);
int musc_match(musc_ctx* ctx, const musc_params* params, musc_hit** hits, uint64_t* nhits);
void musc_free_hits(musc_hit* hits);

int musc_get_stats(musc_ctx* ctx, musc_stats* out);

/* When the last musc_match* left n_overflow_blocks > 0: the (read, window) probes whose
 * (window, key) block may hold more than MaxMatches accepted pairs -- the blocks for which
 * cmd/muscato_confirm/main.go:233-242, 424-448 keep an order-dependent subset.  The library
 * returns every accepted tuple; a host that must reproduce the truncation literally re-derives
 * those blocks from this list (muscato_host.hpp: apply_maxmatches).  Arrays are library-owned
 * until musc_free_u32(); n = 0 when nothing overflowed. */
int musc_overflow_probes(musc_ctx* ctx, uint32_t** read_idx, uint32_t** window, uint64_t* n);
void musc_free_u32(uint32_t* p);

/* ---- several GPUs in one process (a Go host driving one ctx per GPU from locked threads):
 * concatenate the device-resident hits of ctxs[0..n) in rank order into one host array,
 * adding read_base[i] to the read_idx of shard i.  (The one-process-per-GPU path gathers
 * over RCCL instead: muscato_amd/dist.py.) */
int musc_gather(musc_ctx* const* ctxs, int n, const uint64_t* read_base, musc_hit** hits,
                uint64_t* nhits);
/* The same over RCCL: the tuples of ctxs[1..n) travel to ctxs[0]'s GPU over xGMI (ncclSend /
 * ncclRecv, all of them in one ncclGroupStart/End so that the links run side by side), are
 * rebased there and leave the node's GPUs in ONE device-to-host copy -- the concatenation step
 * of the reference (combineWindows, cmd/muscato/main.go:422-505) without N host copies.  One
 * clique per device list is created on first use (ncclCommInitAll) and kept for the process.
 * librccl is loaded at run time on the first call with n > 1: a return code of 20 means it
 * could not be used (the text says why) and musc_gather is the alternative.  The contexts must
 * sit on distinct devices. */
int musc_gather_rccl(musc_ctx* const* ctxs, int n, const uint64_t* read_base, musc_hit** hits,
                     uint64_t* nhits);
/* Whether musc_gather_rccl can load librccl in this process: 0 = yes, 20 = no, with the reason
 * copied into msg (at most cap bytes, NUL-terminated; msg may be NULL).  Needs no GPU.  The library
 * is looked up once per process: MUSC_RCCL_LIB names it, else the ROCm tree's, else the loader's. */
int musc_rccl_probe(char* msg, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* MUSCATO_HIP_H */
