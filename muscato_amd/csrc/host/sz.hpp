// Snappy *framed* streams (.sz), gzip, and whole-file helpers for the muscato host tools.
//
// The reference reads/writes every intermediate and the prepared target database through
// github.com/golang/snappy's framed Reader/Writer (e.g. cmd/muscato_screen/main.go:417,
// cmd/muscato_prep_targets/main.go:246-258).  That module is not in the reference tree and no
// libsnappy exists in this image, so the published framing format
// (https://github.com/google/snappy/blob/main/framing_format.txt) and block format
// (format_description.txt) are implemented here: stream identifier ff 06 00 00 "sNaPpY";
// chunk 0x00 = compressed, 0x01 = uncompressed, both prefixed by a masked CRC-32C of the
// uncompressed bytes; 0x80-0xfd skippable; 0xfe padding; 0x02-0x7f reserved (error).
// The writer emits uncompressed chunks only (always valid, readable by golang/snappy).
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace musc {

inline bool ends_with(const std::string& s, const std::string& suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

inline std::string to_lower(std::string s) {
  for (auto& c : s) c = (char)tolower((unsigned char)c);
  return s;
}

inline std::string slurp(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + path);
  std::string out;
  char buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
  fclose(f);
  return out;
}

inline void spit(const std::string& path, const std::string& data) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot create " + path);
  if (!data.empty() && fwrite(data.data(), 1, data.size(), f) != data.size()) {
    fclose(f);
    throw std::runtime_error("short write to " + path);
  }
  fclose(f);
}

// ---- CRC-32C (Castagnoli), table driven
inline uint32_t crc32c(const uint8_t* p, size_t n) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

inline uint32_t mask_crc(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

// ---- snappy block decode
inline void snappy_block_decode(const uint8_t* p, size_t n, std::string& out) {
  size_t pos = 0;
  uint64_t ulen = 0;
  int shift = 0;
  for (;;) {
    if (pos >= n) throw std::runtime_error("snappy: truncated length");
    const uint8_t b = p[pos++];
    ulen |= (uint64_t)(b & 0x7F) << shift;
    if (b < 0x80) break;
    shift += 7;
    if (shift > 35) throw std::runtime_error("snappy: bad length");
  }
  const size_t base = out.size();
  out.reserve(base + ulen);
  while (pos < n) {
    const uint8_t tag = p[pos++];
    const int t = tag & 3;
    if (t == 0) {
      size_t len = tag >> 2;
      if (len >= 60) {
        const int nb = (int)len - 59;
        if (pos + nb > n) throw std::runtime_error("snappy: truncated literal length");
        len = 0;
        for (int i = 0; i < nb; i++) len |= (size_t)p[pos + i] << (8 * i);
        pos += nb;
      }
      len += 1;
      if (pos + len > n) throw std::runtime_error("snappy: truncated literal");
      out.append((const char*)p + pos, len);
      pos += len;
      continue;
    }
    size_t len, off;
    if (t == 1) {
      if (pos + 1 > n) throw std::runtime_error("snappy: truncated copy");
      len = ((tag >> 2) & 7) + 4;
      off = ((size_t)(tag >> 5) << 8) | p[pos];
      pos += 1;
    } else if (t == 2) {
      if (pos + 2 > n) throw std::runtime_error("snappy: truncated copy");
      len = (tag >> 2) + 1;
      off = p[pos] | ((size_t)p[pos + 1] << 8);
      pos += 2;
    } else {
      if (pos + 4 > n) throw std::runtime_error("snappy: truncated copy");
      len = (tag >> 2) + 1;
      off = p[pos] | ((size_t)p[pos + 1] << 8) | ((size_t)p[pos + 2] << 16) | ((size_t)p[pos + 3] << 24);
      pos += 4;
    }
    if (off == 0 || off > out.size() - base) throw std::runtime_error("snappy: bad copy offset");
    for (size_t i = 0; i < len; i++) out.push_back(out[out.size() - off]);
  }
  if (out.size() - base != ulen) throw std::runtime_error("snappy: length mismatch");
}

inline std::string sz_decode(const std::string& raw) {
  std::string out;
  size_t pos = 0;
  const uint8_t* p = (const uint8_t*)raw.data();
  while (pos < raw.size()) {
    if (pos + 4 > raw.size()) throw std::runtime_error("sz: truncated chunk header");
    const uint8_t type = p[pos];
    const size_t len = p[pos + 1] | ((size_t)p[pos + 2] << 8) | ((size_t)p[pos + 3] << 16);
    pos += 4;
    if (pos + len > raw.size()) throw std::runtime_error("sz: truncated chunk");
    const uint8_t* body = p + pos;
    pos += len;
    if (type == 0xFF) {
      if (len != 6 || memcmp(body, "sNaPpY", 6) != 0) throw std::runtime_error("sz: bad stream identifier");
    } else if (type == 0x00 || type == 0x01) {
      if (len < 4) throw std::runtime_error("sz: chunk too short");
      const uint32_t want = body[0] | ((uint32_t)body[1] << 8) | ((uint32_t)body[2] << 16) | ((uint32_t)body[3] << 24);
      const size_t start = out.size();
      if (type == 0x00) snappy_block_decode(body + 4, len - 4, out);
      else out.append((const char*)body + 4, len - 4);
      if (mask_crc(crc32c((const uint8_t*)out.data() + start, out.size() - start)) != want)
        throw std::runtime_error("sz: checksum mismatch");
    } else if (type >= 0x80) {
      continue;  // skippable / padding
    } else {
      throw std::runtime_error("sz: reserved unskippable chunk");
    }
  }
  return out;
}

inline std::string sz_encode(const std::string& data) {
  std::string out;
  out.append("\xff\x06\x00\x00sNaPpY", 10);
  size_t pos = 0;
  while (pos < data.size()) {
    const size_t n = std::min<size_t>(65536, data.size() - pos);
    const uint32_t c = mask_crc(crc32c((const uint8_t*)data.data() + pos, n));
    const size_t len = n + 4;
    const char hdr[8] = {0x01, (char)(len & 0xFF), (char)((len >> 8) & 0xFF), (char)((len >> 16) & 0xFF),
                         (char)(c & 0xFF), (char)((c >> 8) & 0xFF), (char)((c >> 16) & 0xFF), (char)((c >> 24) & 0xFF)};
    out.append(hdr, 8);
    out.append(data, pos, n);
    pos += n;
  }
  return out;
}

inline std::string gz_decode_file(const std::string& path) {
  gzFile f = gzopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + path);
  std::string out;
  char buf[1 << 16];
  int n;
  while ((n = gzread(f, buf, sizeof buf)) > 0) out.append(buf, n);
  const bool bad = n < 0;
  gzclose(f);
  if (bad) throw std::runtime_error("gzip error in " + path);
  return out;
}

// Read a file, transparently un-snappy-ing ".sz" (as tests/test.go:71-89 does for comparisons
// and as every reference tool does for its inputs).
inline std::string read_maybe_sz(const std::string& path) {
  std::string raw = slurp(path);
  // A framed snappy stream says so itself: stream identifier chunk ff 06 00 00 "sNaPpY".  The
  // reference reads its inputs through snappy.NewReader whatever they are called
  // (cmd/muscato_screen/main.go:417), so the name must not decide; a file named .sz that is not
  // a snappy stream fails in sz_decode, as it fails there.
  static const char magic[10] = {(char)0xff, 0x06, 0x00, 0x00, 's', 'N', 'a', 'P', 'p', 'Y'};
  const bool is_stream = raw.size() >= sizeof magic && memcmp(raw.data(), magic, sizeof magic) == 0;
  if (is_stream || ends_with(to_lower(path), ".sz")) return sz_decode(raw);
  return raw;
}

// Split into lines the way bufio.Scanner(ScanLines) does: '\n' terminated, a trailing "\r" is
// dropped, a final unterminated line is kept, no empty line after a trailing '\n'.
inline std::vector<std::string> split_lines(const std::string& s) {
  std::vector<std::string> out;
  size_t pos = 0;
  while (pos < s.size()) {
    size_t e = s.find('\n', pos);
    size_t end = e == std::string::npos ? s.size() : e;
    size_t len = end - pos;
    if (len && s[pos + len - 1] == '\r') len--;
    out.emplace_back(s, pos, len);
    if (e == std::string::npos) break;
    pos = e + 1;
  }
  return out;
}

}  // namespace musc
