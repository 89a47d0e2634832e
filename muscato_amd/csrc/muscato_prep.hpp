// muscato_prep.hpp -- read prep on the GPU: bytewise sort of the prepared reads and collapse of
// identical sequences.  Included at the end of muscato_hip.hip (one
// translation unit: it uses musc_ctx and the helpers defined there).
//
// Reference: cmd/muscato/main.go sortReads (GNU `sort` of the `seq\tname` lines under LC_ALL=C)
// followed by cmd/muscato_uniqify/main.go:83-135 (adjacent lines with the same sequence become
// one line `seq\tcount\tname1;name2...`).  Only the sequence column is the GPU's business: it
// returns the sort order (ties in input order; the host orders the names of a group, which is
// what comparing the whole `seq\tname` line amounts to) and the group boundaries, and leaves the
// distinct sequences loaded as the context's reads, exactly as musc_reads_load_ascii of the
// first column of reads_sorted.txt.sz would.
//
// Order: bytewise on A < C < G < T < X (ASCII 65 < 67 < 71 < 84 < 88), a proper prefix sorts
// first (the tab that follows it, 9, is below every letter).  A read becomes ceil(maxlen / 21)
// u64 key words of 3-bit codes (0 = past the end, 1..5 = A C G T other), first base most
// significant, so that comparing the words as integers from the first to the last is that
// order; the sort is LSD over the words with a stable radix sort of (word, read number) pairs
// per word (rocPRIM's device radix sort -- a library primitive, not part of the hot path).

#include <rocprim/device/device_radix_sort.hpp>

#define PREP_BASES_PER_WORD 21

DEV uint32_t prep_code(unsigned char ch) {
  switch (ch) {
    case 'A': return 1u;
    case 'C': return 2u;
    case 'G': return 3u;
    case 'T': return 4u;
    default: return 5u;  // prepared reads hold nothing but A C G T X (cmd/muscato_prep_reads/main.go:33-44)
  }
}

// keys[i] = key word `w` of read perm[i]
MUSC_KERNEL __launch_bounds__(256) void k_prep_keys(const unsigned char* __restrict__ s,
                                                   const uint64_t* __restrict__ off,
                                                   const uint32_t* __restrict__ perm, uint64_t n, uint32_t w,
                                                   uint64_t* __restrict__ keys) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t r = perm[i];
    const uint64_t o = off[r];
    const uint64_t len = off[r + 1] - o;
    const uint64_t first = (uint64_t)w * PREP_BASES_PER_WORD;
    uint64_t key = 0;
    for (uint32_t j = 0; j < PREP_BASES_PER_WORD; j++) {
      const uint64_t q = first + j;
      if (q >= len) break;
      key |= (uint64_t)prep_code(s[o + q]) << (60 - 3 * j);
    }
    keys[i] = key;
  }
}

MUSC_KERNEL void k_prep_iota(uint32_t* __restrict__ p, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    p[i] = (uint32_t)i;
}

// head[i] = 1 when the i-th read in sorted order differs from the one before it
MUSC_KERNEL __launch_bounds__(256) void k_prep_heads(const unsigned char* __restrict__ s,
                                                    const uint64_t* __restrict__ off,
                                                    const uint32_t* __restrict__ perm, uint64_t n,
                                                    uint32_t* __restrict__ head) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t h = 1;
    if (i > 0) {
      const uint32_t a = perm[i], b = perm[i - 1];
      const uint64_t oa = off[a], ob = off[b];
      const uint64_t la = off[a + 1] - oa, lb = off[b + 1] - ob;
      if (la == lb) {
        h = 0;
        for (uint64_t q = 0; q < la; q++)
          if (s[oa + q] != s[ob + q]) {
            h = 1;
            break;
          }
      }
    }
    head[i] = h;
  }
}

// ustart[g] = position in sorted order where group g starts (gid = inclusive scan of head, minus 1)
MUSC_KERNEL void k_prep_starts(const uint32_t* __restrict__ head, const uint32_t* __restrict__ incl, uint64_t n,
                              uint32_t* __restrict__ ustart, uint32_t* __restrict__ uhead,
                              const uint32_t* __restrict__ perm) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    if (head[i]) {
      ustart[incl[i] - 1] = (uint32_t)i;
      uhead[incl[i] - 1] = perm[i];  // the input read that represents the group
    }
    if (i == n - 1) ustart[incl[i]] = (uint32_t)n;
  }
}

// 2-bit records of the distinct reads: record g = input read uhead[g] (k_pack_reads with one
// level of indirection)
MUSC_KERNEL void k_prep_pack(const unsigned char* __restrict__ s, const uint64_t* __restrict__ off,
                            const uint32_t* __restrict__ uhead, uint64_t nunique, int rw,
                            uint32_t* __restrict__ rd, uint32_t* __restrict__ rdm, uint32_t* __restrict__ has_x) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t g = t / rw;
  const int j = (int)(t % rw);
  if (g >= nunique) return;
  const uint32_t r = uhead[g];
  const uint64_t o = off[r];
  const uint32_t len = (uint32_t)(off[r + 1] - o);
  if (j == rw - 1) {
    uint32_t anyx = 0;
    for (uint32_t q = 0; q < len; q++) {
      uint32_t isx;
      (void)ascii_code(s[o + q], &isx);
      anyx |= isx;
    }
    rd[t] = (len & 0xFFFFu) | (anyx ? READ_HAS_X : 0u);
    rdm[t] = 0;
    return;
  }
  uint32_t v = 0, mv = 0;
  for (int b = 0; b < 16; b++) {
    const uint32_t q = (uint32_t)j * 16 + b;
    if (q < len) {
      uint32_t isx;
      const uint32_t c = ascii_code(s[o + q], &isx);
      v |= c << (2 * b);
      mv |= isx << (2 * b);
    }
  }
  rd[t] = v;
  rdm[t] = mv;
  if (mv) atomicOr(has_x, 1u);
}

typedef TmpBufs PrepBufs;  // temporaries of musc_reads_sort_unique, released on every exit path

extern "C" int musc_reads_sort_unique(musc_ctx* c, const char* seqs, const uint64_t* offsets, uint64_t nreads,
                                      int on_device, uint32_t** order, uint32_t** ustart, uint64_t* nunique) {
  if (!c) return 1;
  if (!order || !ustart || !nunique) return fail(c, 2, "musc_reads_sort_unique: NULL output pointer");
  *order = *ustart = nullptr;
  *nunique = 0;
  if ((!seqs || !offsets) && nreads) return fail(c, 2, "musc_reads_sort_unique: NULL input");
  HIPCHK(c, hipSetDevice(c->device));
  free_reads(c);
  if (nreads >= 0xFFFFFFF0ull) return fail(c, 2, "too many reads for 32-bit read numbers");
  if (nreads == 0) {
    c->rw = 4;
    *order = (uint32_t*)malloc(4);
    *ustart = (uint32_t*)calloc(1, 4);
    if (!*order || !*ustart) return fail(c, 7, "out of host memory");
    return 0;
  }
  c->ev_used = 0;
  hipEvent_t e0 = pool_event(c), e1 = pool_event(c);
  if (!e0 || !e1) return fail(c, 10, "hipEventCreate failed");
  PrepBufs B;
  const uint64_t n = nreads;
  const unsigned char* d_s = (const unsigned char*)seqs;
  const uint64_t* d_off = offsets;
  uint64_t total = 0;
  if (!on_device) {
    total = offsets[n];
    unsigned char* ds = nullptr;
    uint64_t* doff = nullptr;
    HIPCHK(c, B.alloc(&ds, total + 64));
    HIPCHK(c, B.alloc(&doff, (n + 1) * 8));
    HIPCHK(c, hipMemcpyAsync(ds, seqs, total, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(doff, offsets, (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    d_s = ds;
    d_off = doff;
  }
  HIPCHK(c, hipEventRecord(e0, c->stream));
  HIPCHK(c, hipMemsetAsync(c->counters + 4, 0, 8, c->stream));
  hipLaunchKernelGGL(k_max_len, dim3(std::min(nblk(n, 256), MAX_GRID)), dim3(256), 0, c->stream, d_off, n, c->counters + 4);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->counters + 4, 8, hipMemcpyDeviceToHost, c->stream));
  uint64_t first = 0;
  HIPCHK(c, hipMemcpyAsync(&first, d_off, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint64_t maxlen = c->h_pinned[0];
  if (first != 0) return fail(c, 2, "read offsets[0] must be 0");
  if (maxlen > 65535) return fail(c, 2, "read of %llu bases exceeds the 65535-base record limit", (unsigned long long)maxlen);

  // ---- LSD sort over the key words
  const uint32_t nw = (uint32_t)std::max<uint64_t>((maxlen + PREP_BASES_PER_WORD - 1) / PREP_BASES_PER_WORD, 1);
  uint64_t *k0 = nullptr, *k1 = nullptr;
  uint32_t *p0 = nullptr, *p1 = nullptr;
  HIPCHK(c, B.alloc(&k0, n * 8));
  HIPCHK(c, B.alloc(&k1, n * 8));
  HIPCHK(c, B.alloc(&p0, n * 4));
  HIPCHK(c, B.alloc(&p1, n * 4));
  size_t tmp_bytes = 0;
  HIPCHK(c, rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, p0, p1, (size_t)n, 0u, 63u, c->stream));
  void* tmp = nullptr;
  HIPCHK(c, B.alloc(&tmp, tmp_bytes));
  const dim3 grid(std::min(nblk(n, 256), MAX_GRID));
  hipLaunchKernelGGL(k_prep_iota, grid, dim3(256), 0, c->stream, p0, n);
  HIPCHK(c, hipGetLastError());
  for (uint32_t w = nw; w-- > 0;) {
    hipLaunchKernelGGL(k_prep_keys, grid, dim3(256), 0, c->stream, d_s, d_off, p0, n, w, k0);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, p0, p1, (size_t)n, 0u, 63u, c->stream));
    std::swap(p0, p1);  // p0 = the order after this word
  }

  // ---- groups of identical sequences
  uint32_t *head = nullptr, *incl = nullptr, *stmp = nullptr, *d_ustart = nullptr, *d_uhead = nullptr;
  HIPCHK(c, B.alloc(&head, n * 4));
  HIPCHK(c, B.alloc(&incl, n * 4));
  HIPCHK(c, B.alloc(&stmp, scan_tmp_elems(n) * 4));
  HIPCHK(c, B.alloc(&d_ustart, (n + 1) * 4));
  HIPCHK(c, B.alloc(&d_uhead, n * 4));
  hipLaunchKernelGGL(k_prep_heads, grid, dim3(256), 0, c->stream, d_s, d_off, p0, n, head);
  HIPCHK(c, hipGetLastError());
  int rc = scan_u32(c, head, incl, n, true, stmp);
  if (rc) return rc;
  hipLaunchKernelGGL(k_prep_starts, grid, dim3(256), 0, c->stream, head, incl, n, d_ustart, d_uhead, p0);
  HIPCHK(c, hipGetLastError());
  uint32_t nu32 = 0;
  HIPCHK(c, hipMemcpyAsync(&nu32, incl + (n - 1), 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint64_t nu = nu32;

  // ---- the distinct sequences become the context's reads
  c->nreads = nu;
  c->max_len = (uint32_t)maxlen;
  int rw = (int)((2 * maxlen + 31) / 32) + 1;
  rw = (rw + 3) & ~3;
  if (rw < 4) rw = 4;
  c->rw = rw;
  const uint64_t words = nu * (uint64_t)rw;
  if (words >= (1ull << 32)) return fail(c, 2, "too many read words for one dispatch (reads x record words >= 2^32)");
  HIPCHK(c, hipMalloc((void**)&c->rd, words * 4 + 256));
  HIPCHK(c, hipMalloc((void**)&c->rdm, words * 4 + 256));
  HIPCHK(c, hipMemsetAsync(c->rd + words, 0, 256, c->stream));
  HIPCHK(c, hipMemsetAsync(c->rdm + words, 0, 256, c->stream));
  uint32_t* d_hasx = c->d_flag;
  HIPCHK(c, hipMemsetAsync(d_hasx, 0, 4, c->stream));
  hipLaunchKernelGGL(k_prep_pack, dim3(nblk(words, 256)), dim3(256), 0, c->stream, d_s, d_off, d_uhead, nu, rw, c->rd,
                     c->rdm, d_hasx);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(e1, c->stream));

  // ---- order and group boundaries for the host (names, counts)
  uint32_t* h_order = (uint32_t*)malloc(n * 4);
  uint32_t* h_ustart = (uint32_t*)malloc((nu + 1) * 4);
  if (!h_order || !h_ustart) {
    free(h_order);
    free(h_ustart);
    return fail(c, 7, "out of host memory for the sort order");
  }
  uint32_t hasx = 0;
  hipError_t e = hipMemcpyAsync(h_order, p0, n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(h_ustart, d_ustart, (nu + 1) * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&hasx, d_hasx, 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) {
    free(h_order);
    free(h_ustart);
    return fail(c, 10, "musc_reads_sort_unique: %s", hipGetErrorString(e));
  }
  c->reads_have_x = hasx != 0;
  if (!hasx) {
    (void)hipFree(c->rdm);
    c->rdm = nullptr;
  }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  c->stats.ms_read_prep = ms;  // device time of the last sort + collapse
  *order = h_order;
  *ustart = h_ustart;
  *nunique = nu;
  return 0;
}
