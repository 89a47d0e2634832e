// Host side of the `muscato` drop-in: the reference's utils.Config / flag surface, read and
// target preparation, and the text post-chain to results.txt -- everything around the GPU hot
// path, which is reached only through the C ABI of include/muscato_hip.h.
//
// The reference host is compiled Go; no Go toolchain exists in this image, so the host is C++
// (see INTEGRATION.md for the cgo binding a Go host would use instead).  Every function cites
// the reference code whose behaviour it reproduces (paths relative to kshedden/muscato).
#pragma once

#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <set>
#include <stdexcept>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_set>
#include <vector>

#include "../../../include/muscato_hip.h"
#include "sz.hpp"

namespace musc {

// ------------------------------------------------------------------------------------
// utils.Config (utils/config.go:10-101) + the two additions of this build
// ------------------------------------------------------------------------------------
struct Config {
  std::string ReadFileName, GeneFileName, GeneIdFileName, ResultsFileName;
  std::vector<int> Windows;
  int WindowWidth = 0;
  uint64_t BloomSize = 0;
  int NumHash = 0;
  double PMatch = 0;
  int MinDinuc = 0;
  std::string TempDir, LogDir;
  int MinReadLength = 0, MaxReadLength = 0, MaxMatches = 0, MaxConfirmProcs = 0, MMTol = 0;
  std::string MatchMode;
  int SortPar = 0;
  std::string SortTemp, SortMem;
  bool NoCleanTemp = false, CPUProfile = false;
  // additions (not in the reference): absolute mismatch budget and GPU selection
  int MaxMismatch = -1;  // --MaxMismatch: nmiss budget for every read instead of PMatch
  int GPUs = 1;          // --GPUs: shard the unique reads over this many devices
  int Device = 0;        // --Device: first device ordinal
};

struct Die : std::runtime_error {
  int code;
  Die(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// ---- minimal JSON reader for the flat config object (encoding/json semantics that matter:
// case-insensitive keys, unknown keys ignored, null leaves the field untouched)
struct JsonCursor {
  const std::string& s;
  size_t p = 0;
  explicit JsonCursor(const std::string& t) : s(t) {}
  void ws() { while (p < s.size() && isspace((unsigned char)s[p])) p++; }
  bool eat(char c) { ws(); if (p < s.size() && s[p] == c) { p++; return true; } return false; }
  void need(char c) { if (!eat(c)) throw Die(1, std::string("config: expected '") + c + "' at offset " + std::to_string(p)); }
  std::string str() {
    need('"');
    std::string o;
    while (p < s.size() && s[p] != '"') {
      char c = s[p++];
      if (c == '\\' && p < s.size()) {
        char e = s[p++];
        switch (e) {
          case 'n': o += '\n'; break; case 't': o += '\t'; break; case 'r': o += '\r'; break;
          case 'b': o += '\b'; break; case 'f': o += '\f'; break;
          case 'u': { unsigned v = 0; for (int i = 0; i < 4 && p < s.size(); i++) v = v * 16 + (unsigned)strtol(std::string(1, s[p++]).c_str(), nullptr, 16);
                      if (v < 0x80) o += (char)v; else if (v < 0x800) { o += (char)(0xC0 | (v >> 6)); o += (char)(0x80 | (v & 0x3F)); }
                      else { o += (char)(0xE0 | (v >> 12)); o += (char)(0x80 | ((v >> 6) & 0x3F)); o += (char)(0x80 | (v & 0x3F)); } break; }
          default: o += e;
        }
      } else {
        o += c;
      }
    }
    need('"');
    return o;
  }
  std::string scalar() {  // number / true / false / null as text
    ws();
    size_t b = p;
    while (p < s.size() && (isalnum((unsigned char)s[p]) || s[p] == '-' || s[p] == '+' || s[p] == '.')) p++;
    return s.substr(b, p - b);
  }
};

inline void config_from_json(const std::string& text, Config& c) {
  JsonCursor j(text);
  j.need('{');
  if (j.eat('}')) return;
  do {
    const std::string key = to_lower(j.str());
    j.need(':');
    j.ws();
    if (j.p < text.size() && text[j.p] == '"') {
      const std::string v = j.str();
      if (key == "readfilename") c.ReadFileName = v; else if (key == "genefilename") c.GeneFileName = v;
      else if (key == "geneidfilename") c.GeneIdFileName = v; else if (key == "resultsfilename") c.ResultsFileName = v;
      else if (key == "tempdir") c.TempDir = v; else if (key == "logdir") c.LogDir = v;
      else if (key == "matchmode") c.MatchMode = v; else if (key == "sorttemp") c.SortTemp = v;
      else if (key == "sortmem") c.SortMem = v;
    } else if (j.p < text.size() && text[j.p] == '[') {
      j.need('[');
      std::vector<int> v;
      if (!j.eat(']')) {
        do { v.push_back(atoi(j.scalar().c_str())); } while (j.eat(','));
        j.need(']');
      }
      if (key == "windows") c.Windows = v;
    } else {
      const std::string v = j.scalar();
      if (v == "null" || v.empty()) continue;
      const bool b = v == "true";
      const double d = (v == "true" || v == "false") ? (double)b : atof(v.c_str());
      if (key == "windowwidth") c.WindowWidth = (int)d; else if (key == "bloomsize") c.BloomSize = (uint64_t)d;
      else if (key == "numhash") c.NumHash = (int)d; else if (key == "pmatch") c.PMatch = d;
      else if (key == "mindinuc") c.MinDinuc = (int)d; else if (key == "minreadlength") c.MinReadLength = (int)d;
      else if (key == "maxreadlength") c.MaxReadLength = (int)d; else if (key == "maxmatches") c.MaxMatches = (int)d;
      else if (key == "maxconfirmprocs") c.MaxConfirmProcs = (int)d; else if (key == "mmtol") c.MMTol = (int)d;
      else if (key == "sortpar") c.SortPar = (int)d; else if (key == "nocleantemp") c.NoCleanTemp = b;
      else if (key == "cpuprofile") c.CPUProfile = b; else if (key == "maxmismatch") c.MaxMismatch = (int)d;
      else if (key == "gpus") c.GPUs = (int)d; else if (key == "device") c.Device = (int)d;
    }
  } while (j.eat(','));
  j.need('}');
}

inline std::string json_escape(const std::string& s) {
  std::string o;
  for (char c : s) {
    if (c == '"' || c == '\\') { o += '\\'; o += c; }
    else if (c == '\n') o += "\\n"; else if (c == '\t') o += "\\t";
    else o += c;
  }
  return o;
}

// saveConfig (cmd/muscato/main.go:680-697): the resolved configuration, one JSON object
inline std::string config_to_json(const Config& c) {
  std::string w = "[";
  for (size_t i = 0; i < c.Windows.size(); i++) w += (i ? "," : "") + std::to_string(c.Windows[i]);
  w += "]";
  char buf[64];
  snprintf(buf, sizeof buf, "%.17g", c.PMatch);
  auto S = [](const std::string& k, const std::string& v) { return "\"" + k + "\":\"" + json_escape(v) + "\""; };
  auto N = [](const std::string& k, long long v) { return "\"" + k + "\":" + std::to_string(v); };
  auto B = [](const std::string& k, bool v) { return "\"" + k + "\":" + (v ? "true" : "false"); };
  return "{" + S("ReadFileName", c.ReadFileName) + "," + S("GeneFileName", c.GeneFileName) + "," +
         S("GeneIdFileName", c.GeneIdFileName) + "," + S("ResultsFileName", c.ResultsFileName) + ",\"Windows\":" + w +
         "," + N("WindowWidth", c.WindowWidth) + "," + N("BloomSize", (long long)c.BloomSize) + "," +
         N("NumHash", c.NumHash) + ",\"PMatch\":" + buf + "," + N("MinDinuc", c.MinDinuc) + "," +
         S("TempDir", c.TempDir) + "," + S("LogDir", c.LogDir) + "," + N("MinReadLength", c.MinReadLength) + "," +
         N("MaxReadLength", c.MaxReadLength) + "," + N("MaxMatches", c.MaxMatches) + "," +
         N("MaxConfirmProcs", c.MaxConfirmProcs) + "," + N("MMTol", c.MMTol) + "," + S("MatchMode", c.MatchMode) +
         "," + N("SortPar", c.SortPar) + "," + S("SortTemp", c.SortTemp) + "," + S("SortMem", c.SortMem) + "," +
         B("NoCleanTemp", c.NoCleanTemp) + "," + B("CPUProfile", c.CPUProfile) + "," +
         N("MaxMismatch", c.MaxMismatch) + "," + N("GPUs", c.GPUs) + "," + N("Device", c.Device) + "}\n";
}

// ---- Go `flag` syntax: -name, --name, -name=value, -name value; bools take no value unless
// written -name=value; parsing stops at the first non-flag or at "--".
struct FlagSpec { const char* name; char kind; const char* help; };  // kind: s i f b

inline const std::vector<FlagSpec>& muscato_flags() {
  // cmd/muscato/main.go:708-732 (same names, kinds and help strings) + the additions
  static const std::vector<FlagSpec> f = {
      {"ConfigFileName", 's', "JSON file containing configuration parameters"},
      {"ReadFileName", 's', "Sequencing read file (fastq format)"},
      {"GeneFileName", 's', "Gene file name (processed form)"},
      {"GeneIdFileName", 's', "Gene ID file name (processed form)"},
      {"ResultsFileName", 's', "File name for results"},
      {"Windows", 's', "Starting position of each window"},
      {"WindowWidth", 'i', "Width of each window"},
      {"BloomSize", 'i', "Size of Bloom filter, in bits"},
      {"NumHash", 'i', "Number of hashses"},
      {"PMatch", 'f', "Required proportion of matching positions"},
      {"MinDinuc", 'i', "Minimum number of dinucleotides to check for match"},
      {"TempDir", 's', "Workspace for temporary files"},
      {"MinReadLength", 'i', "Reads shorter than this length are skipped"},
      {"MaxReadLength", 'i', "Reads longer than this length are truncated"},
      {"MaxMatches", 'i', "Return no more than this number of matches per window"},
      {"MaxConfirmProcs", 'i', "Run this number of match confirmation processes concurrently"},
      {"MMTol", 'i', "Number of mismatches allowed above best fit"},
      {"MatchMode", 's', "'first' or 'best' (retain first/best 'MaxMatches' matches meeting criteria)"},
      {"NoCleanTemp", 'b', "Do not delete temporary files from TempDir"},
      {"SortPar", 'i', "Number of parallel sort processes"},
      {"SortTemp", 's', "Directory to use for sort temp files"},
      {"SortMem", 's', "Gnu sort -S parameter"},
      {"CPUProfile", 'b', "Capture CPU profile data"},
      {"MaxMismatch", 'i', "(addition) absolute mismatch budget per read; overrides PMatch when >= 0"},
      {"GPUs", 'i', "(addition) number of GPUs to shard the reads over"},
      {"Device", 'i', "(addition) first GPU ordinal"},
  };
  return f;
}

inline std::string usage(const char* prog, const std::vector<FlagSpec>& flags) {
  std::vector<FlagSpec> v = flags;
  std::sort(v.begin(), v.end(), [](const FlagSpec& a, const FlagSpec& b) { return strcmp(a.name, b.name) < 0; });
  std::string o = std::string("Usage of ") + prog + ":\n";
  for (auto& f : v) {
    const char* ty = f.kind == 's' ? " string" : f.kind == 'i' ? " int" : f.kind == 'f' ? " float" : "";
    o += std::string("  -") + f.name + ty + "\n    \t" + f.help + "\n";
  }
  return o;
}

// returns name -> value for the flags present; bools map to "true"/"false"
inline std::map<std::string, std::string> parse_flags(int argc, char** argv, const std::vector<FlagSpec>& flags,
                                                      std::vector<std::string>* rest) {
  std::map<std::string, std::string> out;
  int i = 1;
  for (; i < argc; i++) {
    std::string a = argv[i];
    if (a.size() < 2 || a[0] != '-') break;
    if (a == "--") { i++; break; }
    size_t b = a[1] == '-' ? 2 : 1;
    std::string name = a.substr(b), val;
    bool hasval = false;
    size_t eq = name.find('=');
    if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); hasval = true; }
    if (name == "help" || name == "h") throw Die(0, usage(argv[0], flags));
    const FlagSpec* spec = nullptr;
    for (auto& f : flags) if (name == f.name) spec = &f;
    if (!spec) throw Die(2, "flag provided but not defined: -" + name + "\n" + usage(argv[0], flags));
    if (spec->kind == 'b') {
      if (!hasval) val = "true";
      if (val != "true" && val != "false" && val != "1" && val != "0" && val != "t" && val != "f" && val != "T" &&
          val != "F" && val != "TRUE" && val != "FALSE" && val != "True" && val != "False")
        throw Die(2, "invalid boolean value \"" + val + "\" for -" + name);
      val = (val[0] == 't' || val[0] == 'T' || val[0] == '1') ? "true" : "false";
    } else {
      if (!hasval) {
        if (i + 1 >= argc) throw Die(2, "flag needs an argument: -" + name);
        val = argv[++i];
      }
      if (spec->kind == 'i') {
        char* e = nullptr;
        strtoll(val.c_str(), &e, 0);
        if (val.empty() || *e) throw Die(2, "invalid value \"" + val + "\" for flag -" + name + ": parse error");
      } else if (spec->kind == 'f') {
        char* e = nullptr;
        strtod(val.c_str(), &e);
        if (val.empty() || *e) throw Die(2, "invalid value \"" + val + "\" for flag -" + name + ": parse error");
      }
    }
    out[name] = val;
  }
  if (rest) for (; i < argc; i++) rest->push_back(argv[i]);
  return out;
}

// handleArgs (cmd/muscato/main.go:708-831): JSON first, then every non-zero flag overrides
inline Config handle_args(int argc, char** argv) {
  auto fl = parse_flags(argc, argv, muscato_flags(), nullptr);
  Config c;
  auto has = [&](const char* k) { return fl.count(k) > 0; };
  auto I = [&](const char* k) { return (int)strtoll(fl[k].c_str(), nullptr, 0); };
  if (has("ConfigFileName") && !fl["ConfigFileName"].empty()) config_from_json(slurp(fl["ConfigFileName"]), c);
  if (has("ReadFileName") && !fl["ReadFileName"].empty()) c.ReadFileName = fl["ReadFileName"];
  if (has("GeneFileName") && !fl["GeneFileName"].empty()) c.GeneFileName = fl["GeneFileName"];
  if (has("GeneIdFileName") && !fl["GeneIdFileName"].empty()) c.GeneIdFileName = fl["GeneIdFileName"];
  if (has("WindowWidth") && I("WindowWidth")) c.WindowWidth = I("WindowWidth");
  if (has("BloomSize") && I("BloomSize")) c.BloomSize = (uint64_t)strtoll(fl["BloomSize"].c_str(), nullptr, 0);
  if (has("NumHash") && I("NumHash")) c.NumHash = I("NumHash");
  if (has("PMatch") && atof(fl["PMatch"].c_str()) != 0) c.PMatch = atof(fl["PMatch"].c_str());
  if (has("MinDinuc") && I("MinDinuc")) c.MinDinuc = I("MinDinuc");
  if (has("TempDir") && !fl["TempDir"].empty()) c.TempDir = fl["TempDir"];
  if (has("MinReadLength") && I("MinReadLength")) c.MinReadLength = I("MinReadLength");
  if (has("MaxReadLength") && I("MaxReadLength")) c.MaxReadLength = I("MaxReadLength");
  if (has("MaxMatches") && I("MaxMatches")) c.MaxMatches = I("MaxMatches");
  if (has("MaxConfirmProcs") && I("MaxConfirmProcs")) c.MaxConfirmProcs = I("MaxConfirmProcs");
  if (has("MatchMode") && !fl["MatchMode"].empty()) c.MatchMode = fl["MatchMode"];
  if (has("MMTol") && I("MMTol")) c.MMTol = I("MMTol");
  if (has("ResultsFileName") && !fl["ResultsFileName"].empty()) c.ResultsFileName = fl["ResultsFileName"];
  if (has("NoCleanTemp") && fl["NoCleanTemp"] == "true") c.NoCleanTemp = true;
  if (has("CPUProfile") && fl["CPUProfile"] == "true") c.CPUProfile = true;
  if (has("SortPar") && I("SortPar")) c.SortPar = I("SortPar");
  if (has("SortMem") && !fl["SortMem"].empty()) c.SortMem = fl["SortMem"];
  if (has("SortTemp") && !fl["SortTemp"].empty()) c.SortTemp = fl["SortTemp"];
  if (has("MaxMismatch")) c.MaxMismatch = I("MaxMismatch");
  if (has("GPUs") && I("GPUs")) c.GPUs = I("GPUs");
  if (has("Device")) c.Device = I("Device");
  if (c.ResultsFileName.empty()) {
    c.ResultsFileName = "results.txt";
    fputs("ResultsFileName not specified, defaulting to 'results.txt'\n", stderr);
  }
  if (has("Windows") && !fl["Windows"].empty()) {
    c.Windows.clear();
    std::string w = fl["Windows"];
    size_t p = 0;
    while (p <= w.size()) {
      size_t e = w.find(',', p);
      std::string tok = w.substr(p, e == std::string::npos ? std::string::npos : e - p);
      char* end = nullptr;
      long v = strtol(tok.c_str(), &end, 10);
      if (tok.empty() || *end) throw Die(1, "Error in handleArgs: bad Windows value \"" + tok + "\"");
      c.Windows.push_back((int)v);
      if (e == std::string::npos) break;
      p = e + 1;
    }
  }
  return c;
}

// checkArgs (cmd/muscato/main.go:833-904): required fields and defaults, same messages
inline void check_args(Config& c) {
  auto need = [](bool ok, const char* what) {
    if (!ok) throw Die(1, std::string("\n") + what + " not provided, run 'muscato --help for more information.\n\n");
  };
  need(!c.ReadFileName.empty(), "ReadFileName");
  need(!c.GeneFileName.empty(), "GeneFileName");
  need(!c.GeneIdFileName.empty(), "GeneIdFileName");
  need(!c.Windows.empty(), "Windows");
  need(c.WindowWidth != 0, "WindowWidth");
  if (c.BloomSize == 0) { fputs("BloomSize not provided, defaulting to 4 billion\n", stderr); c.BloomSize = 4000000000ull; }
  if (c.NumHash == 0) { fputs("NumHash not provided, defaulting to 20\n", stderr); c.NumHash = 20; }
  if (c.PMatch == 0) { fputs("PMatch not provided, defaulting to 1\n", stderr); c.PMatch = 1; }
  if (c.MaxReadLength == 0) throw Die(1, "MaxReadLength not provided, run 'muscato --help for more information.\n\n");
  if (c.MaxMatches == 0) { fputs("MaxMatches not provided, defaulting to 1 million\n", stderr); c.MaxMatches = 1000000; }
  if (c.MaxConfirmProcs == 0) { fputs("MaxConfirmProcs not provided, defaulting to 3\n", stderr); c.MaxConfirmProcs = 3; }
  if (!ends_with(c.ReadFileName, ".fastq"))
    fprintf(stderr, "Warning: %s may not be a fastq file, continuing anyway\n", c.ReadFileName.c_str());
  if (c.MatchMode.empty()) { fputs("MatchMode not provided, defaulting to 'best'\n", stderr); c.MatchMode = "best"; }
  if (c.MatchMode != "best" && c.MatchMode != "first") throw Die(1, "MatchMode must be 'first' or 'best'\n");
  if (c.SortPar == 0) c.SortPar = 8;
  if (c.SortMem.empty()) { fputs("SortMem not provided, defaulting to 50%\n", stderr); c.SortMem = "50%"; }
  if ((int)c.Windows.size() > MUSC_MAX_WINDOWS)
    throw Die(1, "at most " + std::to_string(MUSC_MAX_WINDOWS) + " windows are supported by this build\n");
  if (c.GPUs < 1) c.GPUs = 1;
}

// ------------------------------------------------------------------------------------
// target preparation (cmd/muscato_prep_targets/main.go)
// ------------------------------------------------------------------------------------
inline void subx(std::string& s) {  // :68-80 and cmd/muscato_prep_reads/main.go:33-44
  for (auto& c : s) if (c != 'A' && c != 'T' && c != 'C' && c != 'G') c = 'X';
}

inline std::string revcomp(const std::string& s) {  // :48-66 (bytes outside ATGCX become 0)
  std::string b(s.size(), '\0');
  const size_t m = s.size();
  for (size_t i = 0; i < m; i++) {
    char o = 0;
    switch (s[i]) { case 'A': o = 'T'; break; case 'T': o = 'A'; break; case 'G': o = 'C'; break;
                    case 'C': o = 'G'; break; case 'X': o = 'X'; break; }
    b[m - 1 - i] = o;
  }
  return b;
}

struct PreparedTargets {
  std::vector<std::string> seqs;  // one line each of musc_<file>.sz
  std::vector<std::string> ids;   // "%011d\tname\tlen" lines of musc_ids_<file>.sz
};

inline std::string id_line(size_t num, const std::string& name, size_t len) {
  char b[32];
  snprintf(b, sizeof b, "%011zu", num);
  return std::string(b) + "\t" + name + "\t" + std::to_string(len);
}

inline PreparedTargets prep_targets_text(const std::string& raw, bool rev) {  // processText :82-141
  PreparedTargets out;
  size_t lnum = 0;
  for (auto& line : split_lines(raw)) {
    if (line.empty()) break;
    size_t t = line.find('\t');
    if (t == std::string::npos || line.find('\t', t + 1) != std::string::npos) break;  // reference logs + exit(0)
    std::string nam = line.substr(0, t), seq = line.substr(t + 1);
    subx(seq);
    out.seqs.push_back(seq);
    if (rev) out.seqs.push_back(revcomp(seq));
    out.ids.push_back(id_line(lnum++, nam, seq.size()));
    if (rev) out.ids.push_back(id_line(lnum++, nam + "_r", seq.size()));
  }
  return out;
}

inline PreparedTargets prep_targets_fasta(const std::string& raw, bool rev) {  // processFasta :143-213
  PreparedTargets out;
  std::string name, seq;
  auto flush = [&](const std::string& s, bool r) {
    out.seqs.push_back(s);
    out.ids.push_back(id_line(out.ids.size(), name + (r ? "_r" : ""), s.size()));
  };
  for (auto& line : split_lines(raw)) {
    if (line.empty()) throw Die(1, "muscato_prep_targets: empty line in FASTA input (the reference panics here)");
    if (line[0] == '>') {
      if (!seq.empty()) {
        subx(seq);
        flush(seq, false);
        if (rev) flush(revcomp(seq), true);
      }
      name = line;  // the leading '>' is kept (:199)
      seq.clear();
      continue;
    }
    seq += line;
  }
  if (!seq.empty()) {  // the last record is NOT subx-ed (:204-212)
    flush(seq, false);
    if (rev) flush(revcomp(seq), true);
  }
  return out;
}

// targets() + main (:215-333): .gz/.sz inputs, "fasta" decided on the original file name,
// outputs musc_<file>.sz and musc_ids_<file>.sz next to the input
inline void prep_targets_file(const std::string& path, bool rev, std::string* seq_out, std::string* id_out) {
  const std::string low = to_lower(path);
  std::string raw;
  if (ends_with(low, ".gz")) raw = gz_decode_file(path);
  else if (ends_with(low, ".sz")) raw = sz_decode(slurp(path));
  else raw = slurp(path);
  PreparedTargets pt = ends_with(low, "fasta") ? prep_targets_fasta(raw, rev) : prep_targets_text(raw, rev);
  size_t slash = path.rfind('/');
  std::string dir = slash == std::string::npos ? "" : path.substr(0, slash + 1);
  std::string file = slash == std::string::npos ? path : path.substr(slash + 1);
  std::string lowf = to_lower(file);
  if (ends_with(lowf, ".gz") || ends_with(lowf, ".sz")) file = file.substr(0, file.size() - 3);
  *seq_out = dir + "musc_" + file + ".sz";
  *id_out = dir + "musc_ids_" + file + ".sz";
  std::string a, b;
  for (auto& s : pt.seqs) { a += s; a += '\n'; }
  for (auto& s : pt.ids) { b += s; b += '\n'; }
  spit(*seq_out, sz_encode(a));
  spit(*id_out, sz_encode(b));
}

// Views into a text: the same line and field rules as split_lines / fields, without copies (the
// side outputs walk results.txt three times; at a million lines the copies were most of a run).
template <class F>
inline void for_each_line(const std::string& s, F f) {
  size_t pos = 0;
  while (pos < s.size()) {
    const size_t e = s.find('\n', pos);
    const size_t end = e == std::string::npos ? s.size() : e;
    size_t len = end - pos;
    if (len && s[pos + len - 1] == '\r') len--;
    f(std::string_view(s.data() + pos, len));
    if (e == std::string::npos) break;
    pos = e + 1;
  }
}

inline int fields_sv(std::string_view s, std::string_view* out, int maxf) {  // bytes.Fields, first maxf
  int n = 0;
  size_t i = 0;
  while (i < s.size() && n < maxf) {
    while (i < s.size() && isspace((unsigned char)s[i])) i++;
    const size_t b = i;
    while (i < s.size() && !isspace((unsigned char)s[i])) i++;
    if (i > b) out[n++] = s.substr(b, i - b);
  }
  return n;
}

// ------------------------------------------------------------------------------------
// read preparation: utils/fastq.go, cmd/muscato_prep_reads, `sort`, cmd/muscato_uniqify
// ------------------------------------------------------------------------------------
struct UniqueRead {
  std::string seq;
  size_t count;
  std::string names;
};


// ------------------------------------------------------------------------------------
// post-chain (cmd/muscato/main.go:422-676) and side outputs
// ------------------------------------------------------------------------------------

// Go's strings.Fields / bytes.Fields on ASCII whitespace
inline std::vector<std::string> fields(const std::string& s) {
  std::vector<std::string> f;
  size_t i = 0;
  while (i < s.size()) {
    while (i < s.size() && isspace((unsigned char)s[i])) i++;
    size_t b = i;
    while (i < s.size() && !isspace((unsigned char)s[i])) i++;
    if (i > b) f.emplace_back(s, b, i - b);
  }
  return f;
}

// hits must already be the per-read best+MMTol selection (matches.txt).  Produces the bytes
// of ResultsFileName: sort -k5 + join with the id file + cut (:524-611) turns the gene number
// into "name\tlen" (hits whose number is absent from the id file are unpairable and vanish);
// sort -k1 (:657) orders the six-column lines bytewise; join (:659) appends count and names.
inline std::string results_text(const musc_hit* hits, size_t nhits, const std::vector<UniqueRead>& reads,
                                const std::vector<std::string>& targets,
                                const std::map<uint64_t, std::string>& id_rest) {
  struct Line { std::string six; uint32_t read; };
  std::vector<Line> lines;
  lines.reserve(nhits);
  for (size_t i = 0; i < nhits; i++) {
    const musc_hit& h = hits[i];
    auto it = id_rest.find(h.gene_idx);
    if (it == id_rest.end()) continue;
    const std::string& r = reads[h.read_idx].seq;
    std::string six = r;
    six += '\t';
    six.append(targets[h.gene_idx], h.pos, r.size());
    six += '\t';
    six += std::to_string(h.pos);
    six += '\t';
    six += std::to_string(h.nmiss);
    six += '\t';
    six += it->second;
    lines.push_back(Line{std::move(six), h.read_idx});
  }
  std::sort(lines.begin(), lines.end(), [](const Line& a, const Line& b) { return a.six < b.six; });
  std::string out;
  for (auto& l : lines) {
    out += l.six;
    out += '\t';
    out += std::to_string(reads[l.read].count);
    out += '\t';
    out += reads[l.read].names;
    out += '\n';
  }
  return out;
}

inline std::string nonmatch_name(const std::string& results) {  // cmd/muscato_nonmatch/main.go:67-73
  size_t slash = results.rfind('/');
  std::string a = slash == std::string::npos ? "" : results.substr(0, slash + 1);
  std::string b = slash == std::string::npos ? results : results.substr(slash + 1);
  std::vector<std::string> c;
  size_t p = 0;
  for (;;) {
    size_t e = b.find('.', p);
    c.push_back(b.substr(p, e == std::string::npos ? std::string::npos : e - p));
    if (e == std::string::npos) break;
    p = e + 1;
  }
  std::string d = c.back();
  c.back() = "nonmatch";
  c.push_back(d + ".fastq");
  std::string j;
  for (size_t i = 0; i < c.size(); i++) { if (i) j += '.'; j += c[i]; }
  return a + j;
}

// cmd/muscato_nonmatch/main.go:95-114 with an exact set in place of the Bloom filter
inline std::string nonmatch_text(const std::string& results, const std::vector<UniqueRead>& reads) {
  std::unordered_set<std::string_view> matched;
  for_each_line(results, [&](std::string_view line) {
    std::string_view f[1];
    if (fields_sv(line, f, 1)) matched.insert(f[0]);
  });
  std::string out;
  for (auto& u : reads) {
    if (matched.count(std::string_view(u.seq))) continue;
    auto f = fields(u.seq + "\t" + std::to_string(u.count) + "\t" + u.names);
    if (f.size() < 3) continue;  // the reference would panic on an empty name
    out += f[2] + "#" + f[1] + "\n" + f[0] + "\n+\n" + std::string(f[0].size(), '!') + "\n";
  }
  return out;
}

inline std::string stats_name(const std::string& results, const char* tag) {  // path.Ext handling, :113-120
  size_t slash = results.rfind('/');
  size_t dot = results.rfind('.');
  if (dot != std::string::npos && (slash == std::string::npos || dot > slash))
    return results.substr(0, dot) + tag + results.substr(dot);
  return results + tag;
}

// cmd/muscato/main.go:94-150 + cmd/muscato_genestats/main.go: sort -k5, count runs of column 5
inline std::string genestats_text(const std::string& results) {
  struct Ln {
    std::string_view line, key;  // key: GNU sort -k5 = from the blank before field 5 to the end
  };
  std::vector<Ln> lines;
  for_each_line(results, [&](std::string_view l) {
    size_t p = 0;
    for (int f = 0; f < 4; f++) {
      while (p < l.size() && (l[p] == ' ' || l[p] == '\t')) p++;
      while (p < l.size() && l[p] != ' ' && l[p] != '\t') p++;
    }
    lines.push_back(Ln{l, l.substr(p)});
  });
  std::sort(lines.begin(), lines.end(), [](const Ln& a, const Ln& b) {
    const int c = a.key.compare(b.key);
    if (c) return c < 0;
    return a.line < b.line;
  });
  std::string out;
  std::string_view old;
  size_t n = 0;
  bool first = true;
  auto flush = [&]() {
    out.append(old.data(), old.size());
    out += "\t" + std::to_string(n) + "\t\n";
  };
  for (auto& l : lines) {
    std::string_view f[5];
    if (fields_sv(l.line, f, 5) < 5) continue;
    if (first) { old = f[4]; first = false; }
    if (f[4] != old) { flush(); old = f[4]; n = 0; }
    n++;
  }
  if (!first) flush();
  return out;
}

// cmd/muscato_readstats/main.go: per run of equal names-column tokens, the set of gene names.
// The reference prints the set in Go map order (random); here it is sorted.
inline std::string readstats_text(const std::string& results) {
  std::string out;
  std::string_view old;
  std::set<std::string_view> genes;
  bool first = true;
  auto flush = [&]() {
    out.append(old.data(), old.size());
    out += '\t';
    for (auto& g : genes) { out.append(g.data(), g.size()); out += ';'; }
    out += '\n';
  };
  for_each_line(results, [&](std::string_view l) {
    std::string_view f[8];
    if (fields_sv(l, f, 8) < 8) return;
    if (first) { old = f[7]; first = false; }
    if (f[7] != old) { flush(); old = f[7]; genes.clear(); }
    genes.insert(f[4]);
  });
  if (!first) flush();
  return out;
}

// ------------------------------------------------------------------------------------
// the pipeline (cmd/muscato/main.go:1005-1058), hot path through the C ABI
// ------------------------------------------------------------------------------------
inline void mkdir_p(const std::string& path) {
  std::string cur;
  for (size_t i = 0; i <= path.size(); i++) {
    if (i == path.size() || path[i] == '/') {
      if (!cur.empty() && mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST)
        throw Die(1, "Directory " + cur + " does not exist and cannot be created.");
    }
    if (i < path.size()) cur += path[i];
  }
}

inline std::string join_path(const std::string& a, const std::string& b) {
  if (a.empty()) return b;
  return a.back() == '/' ? a + b : a + "/" + b;
}

inline std::string make_uid() {  // stands in for uuid.NewUUID (cmd/muscato/main.go:933)
  std::random_device rd;
  char b[40];
  snprintf(b, sizeof b, "%08x-%04x-%04x-%04x-%08x%04x", (unsigned)rd(), (unsigned)rd() & 0xFFFF, (unsigned)rd() & 0xFFFF,
           (unsigned)rd() & 0xFFFF, (unsigned)rd(), (unsigned)rd() & 0xFFFF);
  return b;
}

struct Logger {
  FILE* f = nullptr;
  void open(const std::string& path) { f = fopen(path.c_str(), "w"); }
  void printf(const char* fmt, ...) {
    if (!f) return;
    time_t t = time(nullptr);
    struct tm tmv;
    localtime_r(&t, &tmv);
    fprintf(f, "%02d:%02d:%02d ", tmv.tm_hour, tmv.tm_min, tmv.tm_sec);
    va_list ap;
    va_start(ap, fmt);
    vfprintf(f, fmt, ap);
    va_end(ap);
    fputc('\n', f);
    fflush(f);
  }
  ~Logger() { if (f) fclose(f); }
};

// wall time of the stages of a run, written to muscato.log (where the time goes end to end)
struct StageClock {
  Logger& log;
  double t0, last;
  static double now() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
  }
  explicit StageClock(Logger& l) : log(l), t0(now()), last(t0) {}
  void lap(const char* what) {
    const double t = now();
    log.printf("stage %-28s %8.3f s (total %.3f s)", what, t - last, t - t0);
    last = t;
  }
};

inline musc_params to_params(const Config& c) {
  musc_params p;
  memset(&p, 0, sizeof p);
  p.n_windows = (int)c.Windows.size();
  for (int i = 0; i < p.n_windows; i++) p.windows[i] = c.Windows[i];
  p.window_width = c.WindowWidth;
  p.pmatch = c.PMatch;
  p.min_dinuc = c.MinDinuc;
  p.max_read_length = c.MaxReadLength;
  p.max_matches = c.MaxMatches;
  p.match_mode = c.MatchMode == "first" ? 1 : 0;
  p.mmtol = c.MMTol;
  p.apply_mmtol = 1;
  p.max_mismatch_p1 = c.MaxMismatch >= 0 ? c.MaxMismatch + 1 : 0;
  p.n_shards = c.GPUs > 1 ? c.GPUs : 0;
  return p;
}

struct Concat {
  std::string buf;
  std::vector<uint64_t> off;
};

template <class It, class F>
inline Concat concat(It b, It e, F seq_of) {
  Concat c;
  c.off.push_back(0);
  for (It i = b; i != e; ++i) {
    c.buf += seq_of(*i);
    c.off.push_back(c.buf.size());
  }
  c.buf.append(16, '\0');
  return c;
}

// utils/entropy.go:5-40 on ASCII
inline int count_dinuc(const char* seq, int n) {
  bool seen[25] = {false};
  int last = 0, cnt = 0;
  for (int i = 0; i < n; i++) {
    int v;
    switch (seq[i]) { case 'A': v = 0; break; case 'T': v = 1; break; case 'G': v = 2; break; case 'C': v = 3; break; default: v = 4; }
    if (i > 0 && !seen[5 * last + v]) { seen[5 * last + v] = true; cnt++; }
    last = v;
  }
  return cnt;
}

// The reference truncates a (window, key) block of muscato_confirm at MaxMatches accepted pairs
// in an order-dependent way (cmd/muscato_confirm/main.go:183-244: candidates in the bytewise
// order of their smatch_k line outside, reads in the bytewise order of their win_k_sorted line
// inside; "first": append and stop once len > MaxMatches; "best": qinsert :424-448, a sift-up
// heap whose array tail is cut at MaxMatches).  The GPU path returns every accepted tuple and
// says which probes sit in blocks that may overflow; this re-derives those blocks from the
// tuples and replays the truncation literally, then rebuilds the union over windows
// (cmd/muscato/main.go:441-463) for the affected reads.  `all` = every accepted tuple
// (apply_mmtol = 0).  Returns the corrected set of all accepted tuples.
inline std::vector<musc_hit> apply_maxmatches(const Config& cfg, const std::vector<UniqueRead>& reads,
                                              const std::vector<std::string>& targets, std::vector<musc_hit> all,
                                              const uint32_t* probe_read, const uint32_t* probe_win, uint64_t nprobes,
                                              size_t* n_truncated_blocks) {
  const int ww = cfg.WindowWidth, W = (int)cfg.Windows.size();
  if (n_truncated_blocks) *n_truncated_blocks = 0;
  auto valid = [&](uint32_t r, int k) {
    const std::string& s = reads[r].seq;
    const int q1 = cfg.Windows[k], q2 = q1 + ww;
    return (int)s.size() >= q2 && count_dinuc(s.data() + q1, ww) >= cfg.MinDinuc;
  };
  // would window k's confirm emit tuple h (D2-D4 of SURVEY.md 8a)
  auto emits = [&](const musc_hit& h, int k) {
    if (!valid(h.read_idx, k)) return false;
    const std::string& s = reads[h.read_idx].seq;
    const std::string& t = targets[h.gene_idx];
    const int64_t q1 = cfg.Windows[k], jx = (int64_t)h.pos + q1, L = (int64_t)s.size(), T = (int64_t)t.size();
    if (jx + ww > T || memcmp(s.data() + q1, t.data() + jx, ww) != 0) return false;
    if (jx == 0) return L <= std::min<int64_t>(100 - ww, T);
    return (int64_t)h.pos + L <= T;
  };
  std::sort(all.begin(), all.end(), [](const musc_hit& a, const musc_hit& b) {
    if (a.read_idx != b.read_idx) return a.read_idx < b.read_idx;
    if (a.gene_idx != b.gene_idx) return a.gene_idx < b.gene_idx;
    return a.pos < b.pos;
  });
  std::vector<size_t> first_of(reads.size() + 1, all.size());
  for (size_t i = all.size(); i-- > 0;) first_of[all[i].read_idx] = i;
  for (size_t r = reads.size(); r-- > 0;) if (first_of[r] == all.size()) first_of[r] = first_of[r + 1];

  // candidate blocks named by the probes
  typedef std::pair<int, std::string> BlockId;
  std::map<BlockId, std::vector<uint32_t>> block_reads;
  for (uint64_t i = 0; i < nprobes; i++) {
    const int k = (int)probe_win[i];
    if (k >= W || probe_read[i] >= reads.size() || !valid(probe_read[i], k)) continue;
    block_reads[BlockId(k, reads[probe_read[i]].seq.substr(cfg.Windows[k], ww))];
  }
  if (block_reads.empty()) return all;
  for (uint32_t r = 0; r < reads.size(); r++)
    for (int k = 0; k < W; k++) {
      if (!valid(r, k)) continue;
      auto it = block_reads.find(BlockId(k, reads[r].seq.substr(cfg.Windows[k], ww)));
      if (it != block_reads.end()) it->second.push_back(r);
    }

  struct Key3 { uint32_t r, g, p; bool operator<(const Key3& o) const { return r != o.r ? r < o.r : g != o.g ? g < o.g : p < o.p; } };
  std::map<BlockId, std::set<Key3>> kept;  // only for blocks that really overflow
  for (auto& br : block_reads) {
    const int k = br.first.first, q1 = cfg.Windows[k], q2 = q1 + ww;
    std::vector<uint32_t>& R = br.second;
    // reads in the order of their "key \t left \t right" line (cmd/muscato/main.go:261-270)
    std::sort(R.begin(), R.end(), [&](uint32_t a, uint32_t b) {
      const std::string &x = reads[a].seq, &y = reads[b].seq;
      return x.substr(0, q1) + "\t" + x.substr(q2) < y.substr(0, q1) + "\t" + y.substr(q2);
    });
    // accepted pairs of this block, grouped by candidate (gene, window start)
    std::map<std::pair<uint32_t, uint32_t>, std::map<uint32_t, uint32_t>> by_cand;  // cand -> read -> nx
    size_t npairs = 0;
    for (uint32_t r : R)
      for (size_t i = first_of[r]; i < first_of[r + 1]; i++)
        if (emits(all[i], k)) {
          by_cand[std::make_pair(all[i].gene_idx, all[i].pos + (uint32_t)q1)][r] = all[i].nmiss;
          npairs++;
        }
    if ((int64_t)npairs <= (int64_t)cfg.MaxMatches) continue;  // a hash-collision false alarm
    if (n_truncated_blocks) ++*n_truncated_blocks;
    // candidates in the order of their "key \t left \t right \t %011d \t pos" line
    // (cmd/muscato_screen/main.go:303-316, 341-363; cmd/muscato/main.go:341-350)
    struct Cand { std::string line; uint32_t g, jx; };
    std::vector<Cand> C;
    for (auto& c : by_cand) {
      const std::string& t = targets[c.first.first];
      const int64_t jx = c.first.second, T = (int64_t)t.size();
      std::string left, right;
      if (jx == 0) {
        right = t.substr(ww, std::max<int64_t>(std::min<int64_t>(100 - q2, T) - ww, 0));
      } else {
        left = t.substr(jx - q1, q1);
        const int64_t jy = jx + ww, jz = std::min<int64_t>(jy + cfg.MaxReadLength - q2, T);
        right = t.substr(jy, std::max<int64_t>(jz - jy, 0));
      }
      char gid[16];
      snprintf(gid, sizeof gid, "%011u", c.first.first);
      C.push_back(Cand{left + "\t" + right + "\t" + gid + "\t" + std::to_string(jx), c.first.first, c.first.second});
    }
    std::sort(C.begin(), C.end(), [](const Cand& a, const Cand& b) { return a.line < b.line; });
    struct Q { int mm; Key3 t; };
    std::vector<Q> q;
    const bool first = cfg.MatchMode == "first";
    bool stop = false;
    for (auto& c : C) {
      const auto& acc = by_cand[std::make_pair(c.g, c.jx)];
      for (uint32_t r : R) {
        auto it = acc.find(r);
        if (it == acc.end()) continue;
        Q qq{(int)it->second, Key3{r, c.g, c.jx - (uint32_t)q1}};
        if (first) {
          q.push_back(qq);
          if ((int64_t)q.size() > (int64_t)cfg.MaxMatches) { stop = true; break; }  // :236-238
        } else {
          q.push_back(qq);  // qinsert :424-448
          size_t ii = q.size() - 1;
          while (ii > 0) {
            const size_t jj = (ii - 1) / 2;
            if (q[jj].mm > q[ii].mm) { std::swap(q[jj], q[ii]); ii = jj; } else break;
          }
          if ((int64_t)q.size() > (int64_t)cfg.MaxMatches) q.resize(cfg.MaxMatches);
        }
      }
      if (stop) break;
    }
    std::set<Key3>& ks = kept[br.first];
    for (auto& e : q) ks.insert(e.t);
  }
  if (kept.empty()) return all;

  // rebuild the union for every read that has a window in a truncated block
  std::vector<char> affected(reads.size(), 0);
  for (auto& kb : kept)
    for (uint32_t r : block_reads[kb.first]) affected[r] = 1;
  std::vector<musc_hit> out;
  out.reserve(all.size());
  for (const musc_hit& h : all) {
    if (!affected[h.read_idx]) { out.push_back(h); continue; }
    bool survive = false;
    for (int k = 0; k < W && !survive; k++) {
      if (!emits(h, k)) continue;
      auto it = kept.find(BlockId(k, reads[h.read_idx].seq.substr(cfg.Windows[k], ww)));
      survive = it == kept.end() || it->second.count(Key3{h.read_idx, h.gene_idx, h.pos});
    }
    if (survive) out.push_back(h);
  }
  return out;
}

// per read keep nmiss <= best + MMTol (cmd/muscato_combine_windows/main.go:36-60)
inline std::vector<musc_hit> best_filter(const std::vector<musc_hit>& all, size_t nreads, int mmtol) {
  std::vector<uint32_t> best(nreads, 0xFFFFFFFFu);
  for (auto& h : all) best[h.read_idx] = std::min(best[h.read_idx], h.nmiss);
  std::vector<musc_hit> out;
  for (auto& h : all) if (h.nmiss <= best[h.read_idx] + (uint32_t)mmtol) out.push_back(h);
  return out;
}

// Load the targets, shard the unique reads over cfg.GPUs devices (one host thread + one
// musc_ctx per device, as the ABI's threading rule asks), run the hot path, gather.
inline std::vector<musc_hit> run_hot_path(const Config& cfg, const std::vector<UniqueRead>& reads,
                                          const std::vector<std::string>& targets, Logger& log, musc_stats* stats0) {
  const int G = cfg.GPUs;
  const Concat db = concat(targets.begin(), targets.end(), [](const std::string& s) -> const std::string& { return s; });
  if (targets.size() >= 0xFFFFFFFFull) throw Die(1, "too many targets");
  std::vector<musc_ctx*> ctxs(G, nullptr);
  std::vector<uint64_t> base(G, 0);
  std::vector<std::string> errs(G);
  const musc_params P = to_params(cfg);
  std::vector<std::thread> th;
  for (int g = 0; g < G; g++) {
    const size_t lo = reads.size() * (size_t)g / G, hi = reads.size() * (size_t)(g + 1) / G;
    base[g] = lo;
    th.emplace_back([&, g, lo, hi] {
      musc_ctx* c = nullptr;
      if (musc_init(cfg.Device + g, &c)) { errs[g] = musc_last_error(nullptr); return; }
      ctxs[g] = c;
      const Concat rd = concat(reads.begin() + lo, reads.begin() + hi, [](const UniqueRead& u) -> const std::string& { return u.seq; });
      uint64_t n = 0;
      if (musc_db_load_ascii(c, db.buf.data(), db.off.data(), (uint32_t)targets.size(), 0) ||
          musc_reads_load_ascii(c, rd.buf.data(), rd.off.data(), hi - lo, 0) || musc_match_device(c, &P, &n))
        errs[g] = musc_last_error(c);
    });
  }
  for (auto& t : th) t.join();
  std::string err;
  for (int g = 0; g < G; g++) if (!errs[g].empty()) err += "GPU " + std::to_string(cfg.Device + g) + ": " + errs[g] + "\n";
  std::vector<musc_hit> out;
  if (err.empty()) {
    musc_hit* h = nullptr;
    uint64_t n = 0;
    // Every GPU copies its tuples to the host (musc_gather).  MUSC_GATHER=rccl (opt-in until it has run
    // on a multi-GPU node) makes the shards' tuples meet on the first GPU over RCCL/xGMI and leave it
    // in one copy; whatever goes wrong there, the host path takes over.
    const char* genv = getenv("MUSC_GATHER");
    const bool want_rccl = G > 1 && genv && !strcmp(genv, "rccl");
    int grc = want_rccl ? musc_gather_rccl(ctxs.data(), G, base.data(), &h, &n) : musc_gather(ctxs.data(), G, base.data(), &h, &n);
    if (want_rccl && grc != 0) {
      log.printf("RCCL gather failed (%d: %s): gathering through the host", grc, musc_last_error(ctxs[0]));
      grc = musc_gather(ctxs.data(), G, base.data(), &h, &n);
    } else if (want_rccl) {
      log.printf("tuples of %d GPUs gathered on GPU %d over RCCL", G, cfg.Device);
    }
    if (grc) err = musc_last_error(ctxs[0]);
    else {
      out.assign(h, h + n);
      musc_free_hits(h);
    }
    if (stats0) musc_get_stats(ctxs[0], stats0);
    std::string prof = "[";
    for (int g = 0; g < G; g++) {
      musc_stats s;
      musc_get_stats(ctxs[g], &s);
      char pb[1024];
      snprintf(pb, sizeof pb,
               "%s{\"gpu\":%d,\"index_kind\":%u,\"index_bytes\":%llu,\"reads\":%llu,\"read_windows\":%llu,\"index_entries_walked\":%llu,"
               "\"pairs_compared\":%llu,\"accepted\":%llu,\"tuples\":%llu,\"batches\":%u,\"ms_total\":%.4f,\"ms_screen_or_match\":%.4f,"
               "\"ms_confirm\":%.4f,\"ms_scan\":%.4f,\"ms_compact\":%.4f,\"ms_index_build\":%.3f,\"confirm_bytes\":%llu,\"match_bytes\":%llu}",
               g ? "," : "", cfg.Device + g, s.index_kind, (unsigned long long)s.index_bytes, (unsigned long long)s.n_reads,
               (unsigned long long)s.n_read_windows, (unsigned long long)s.n_candidates, (unsigned long long)s.n_pairs,
               (unsigned long long)s.n_accepted, (unsigned long long)s.n_hits, s.n_batches, s.ms_total, s.ms_screen, s.ms_confirm,
               s.ms_scan, s.ms_select, s.ms_index_build, (unsigned long long)s.confirm_bytes, (unsigned long long)s.match_bytes);
      prof += pb;
      log.printf("gpu %d: reads %llu windows %llu candidates %llu pairs %llu accepted %llu hits %llu; device %.3f ms "
                 "(screen %.3f scan %.3f confirm %.3f select %.3f), index build %.1f ms, confirm %.1f GB/s",
                 cfg.Device + g, (unsigned long long)s.n_reads, (unsigned long long)s.n_read_windows,
                 (unsigned long long)s.n_candidates, (unsigned long long)s.n_pairs, (unsigned long long)s.n_accepted,
                 (unsigned long long)s.n_hits, s.ms_total, s.ms_screen, s.ms_scan, s.ms_confirm, s.ms_select,
                 s.ms_index_build, s.ms_confirm > 0 ? s.confirm_bytes / 1e6 / s.ms_confirm : 0.0);
    }
    // --CPUProfile (cmd/muscato_screen/main.go:530-538 writes a pprof CPU profile of the screen to
    // LogDir): here the hot path runs on the GPU, so the profile is the per-kernel device timing
    if (cfg.CPUProfile) spit(join_path(cfg.LogDir, "muscato_gpu_profile.json"), prof + "]\n");
  }
  bool overflow = false;
  for (int g = 0; g < G && err.empty(); g++) {
    musc_stats s;
    musc_get_stats(ctxs[g], &s);
    overflow = overflow || (s.n_overflow_blocks != 0 && s.n_overflow_blocks != ~0ull);
  }
  for (int g = 1; g < G; g++) if (ctxs[g]) { musc_destroy(ctxs[g]); ctxs[g] = nullptr; }
  if (err.empty() && overflow) {
    // Some (window,key) block may exceed MaxMatches: redo the pass on one GPU with every read,
    // take all accepted tuples and replay the reference's truncation on the host.
    fputs("MaxMatches reached in at least one window-key block: replaying the reference's truncation...\n", stderr);
    musc_ctx* c = ctxs[0];
    musc_params P1 = P;
    P1.apply_mmtol = 0;
    P1.n_shards = 0;
    const Concat rd = concat(reads.begin(), reads.end(), [](const UniqueRead& u) -> const std::string& { return u.seq; });
    musc_hit* h = nullptr;
    uint64_t n = 0, np = 0;
    uint32_t *pr = nullptr, *pw = nullptr;
    if (musc_reads_load_ascii(c, rd.buf.data(), rd.off.data(), reads.size(), 0) || musc_match(c, &P1, &h, &n) ||
        musc_overflow_probes(c, &pr, &pw, &np)) {
      err = musc_last_error(c);
    } else {
      size_t ntrunc = 0;
      std::vector<musc_hit> all(h, h + n);
      all = apply_maxmatches(cfg, reads, targets, std::move(all), pr, pw, np, &ntrunc);
      out = best_filter(all, reads.size(), cfg.MMTol);
      log.printf("MaxMatches: %llu suspect probes, %zu blocks truncated as the reference does", (unsigned long long)np, ntrunc);
      if (stats0) stats0->n_overflow_blocks = 0;  // handled exactly
    }
    musc_free_hits(h);
    musc_free_u32(pr);
    musc_free_u32(pw);
  }
  for (auto c : ctxs) if (c) musc_destroy(c);
  if (!err.empty()) throw Die(1, "muscato hot path failed:\n" + err);
  return out;
}

// utils/fastq.go + cmd/muscato_prep_reads on the host (parse, MinReadLength on the raw length,
// subx, MaxReadLength, the 1000-character name rule); the `sort` of the `seq\tname` lines and
// cmd/muscato_uniqify/main.go:83-135 through musc_reads_sort_unique: the device orders the
// sequences bytewise and groups the identical ones, the host orders each group's names
// (= comparing the whole `seq\tname` line) and joins them.
inline std::vector<UniqueRead> prep_reads(const std::string& fastq, const Config& c, size_t* n_total) {
  std::vector<std::string> seqs, names;
  {  // utils/fastq.go:35-61: records of four lines, name = whole first line, an incomplete last one is dropped
    std::string_view rec[4];
    int k = 0;
    for_each_line(fastq, [&](std::string_view line) {
      rec[k++] = line;
      if (k < 4) return;
      k = 0;
      if ((int)rec[1].size() < c.MinReadLength) return;
      std::string seq(rec[1]);
      subx(seq);
      if ((int)seq.size() > c.MaxReadLength) seq.resize(c.MaxReadLength);
      std::string rn(rec[0]);
      if (rn.size() > 1000) rn = rn.substr(0, 995) + "...";
      seqs.push_back(std::move(seq));
      names.push_back(std::move(rn));
    });
  }
  if (n_total) *n_total = seqs.size();
  std::vector<UniqueRead> out;
  if (seqs.empty()) return out;
  const Concat rd = concat(seqs.begin(), seqs.end(), [](const std::string& s) -> const std::string& { return s; });
  musc_ctx* ctx = nullptr;
  if (musc_init(c.Device, &ctx)) throw Die(1, std::string("read prep: ") + musc_last_error(nullptr));
  uint32_t *order = nullptr, *ustart = nullptr;
  uint64_t nu = 0;
  if (musc_reads_sort_unique(ctx, rd.buf.data(), rd.off.data(), seqs.size(), 0, &order, &ustart, &nu)) {
    const std::string e = musc_last_error(ctx);
    musc_destroy(ctx);
    throw Die(1, "read prep: " + e);
  }
  musc_destroy(ctx);
  out.reserve(nu);
  std::vector<const std::string*> grp;
  for (uint64_t g = 0; g < nu; g++) {
    grp.clear();
    for (uint32_t k = ustart[g]; k < ustart[g + 1]; k++) grp.push_back(&names[order[k]]);
    std::sort(grp.begin(), grp.end(), [](const std::string* a, const std::string* b) { return *a < *b; });
    std::string na;
    for (size_t i = 0; i < grp.size(); i++) {
      if (i) na += ';';
      na += grp[i]->substr(0, grp[i]->find('\t'));  // toks[1]: what follows a second tab is dropped
    }
    if (na.size() > 1000) na = na.substr(0, 996) + "...";  // cmd/muscato_uniqify/main.go:89-93
    out.push_back(UniqueRead{seqs[order[ustart[g]]], grp.size(), na});
  }
  musc_free_u32(order);
  musc_free_u32(ustart);
  return out;
}

inline int run_muscato(Config cfg) {
  // setupEnvs/makeTemp/setupLog/saveConfig (cmd/muscato/main.go:906-967, 680-706)
  const std::string uid = make_uid();
  cfg.TempDir = join_path(cfg.TempDir.empty() ? "muscato_tmp" : cfg.TempDir, uid);
  mkdir_p(cfg.TempDir);
  cfg.LogDir = join_path(cfg.LogDir.empty() ? "muscato_logs" : cfg.LogDir, uid);
  mkdir_p(cfg.LogDir);
  Logger log;
  log.open(join_path(cfg.LogDir, "muscato.log"));
  spit(join_path(cfg.LogDir, "config.json"), config_to_json(cfg));
  // cleanTmp is deferred in the reference (cmd/muscato/main.go:969-979, 1019): the temporary
  // directory goes away on every exit path, also when a stage fails, unless NoCleanTemp is set
  struct TmpGuard {
    const Config& c;
    ~TmpGuard() {
      if (c.NoCleanTemp) return;
      unlink(join_path(c.TempDir, "reads_sorted.txt.sz").c_str());
      rmdir(c.TempDir.c_str());
    }
  } tmp_guard{cfg};

  StageClock clk(log);
  fputs("Preparing reads...\n", stderr);
  size_t n_total = 0;
  std::vector<UniqueRead> reads = prep_reads(slurp(cfg.ReadFileName), cfg, &n_total);
  if (reads.empty()) throw Die(1, "muscato_uniqify: no input from -");
  fprintf(stderr, "Found %zu total sequences\nFound %zu unique sequences\n", n_total, reads.size());
  spit(join_path(cfg.LogDir, "seqinfo.json"),
       "{\"NumUnique\":" + std::to_string(reads.size()) + ",\"NumTotal\":" + std::to_string(n_total) + "}\n");
  if (cfg.NoCleanTemp) {  // keep the one intermediate other tools consume
    std::string t;
    for (auto& u : reads) t += u.seq + "\t" + std::to_string(u.count) + "\t" + u.names + "\n";
    spit(join_path(cfg.TempDir, "reads_sorted.txt.sz"), sz_encode(t));
  }

  clk.lap("read prep");
  fputs("Windowing reads...\n", stderr);
  for (size_t k = 0; k < cfg.Windows.size(); k++) {  // cmd/muscato_window_reads/main.go:143-151
    size_t nvalid = 0;
    for (auto& u : reads) nvalid += (int)u.seq.size() >= cfg.Windows[k] + cfg.WindowWidth;
    log.printf("Window %zu produced %zu valid reads", k, nvalid);
    if (nvalid == 0) throw Die(1, "Window " + std::to_string(k) + " produced no valid reads, exiting");
  }

  // gene file: sequence = text before the first tab, gene number = line index
  // (cmd/muscato_screen/main.go:439-452); id file: "%011d\tname\tlen" (join field 1)
  std::vector<std::string> targets;
  for (auto& l : split_lines(read_maybe_sz(cfg.GeneFileName))) targets.push_back(l.substr(0, l.find('\t')));
  {
    // Targets are compared as 2-bit bases + an "X" plane: every byte that is not A, C, G or T is an
    // X here.  muscato_prep_targets writes nothing else (it leaves the LAST FASTA record un-substituted,
    // cmd/muscato_prep_targets/main.go:204-212), but a hand-made gene file may: the reference then
    // compares the raw byte, so an 'N' in a target would mismatch an 'X' in a read where this tool
    // counts a match.  Say so instead of differing silently.
    size_t nodd = 0, first_t = 0;
    for (size_t t = 0; t < targets.size(); t++)
      for (unsigned char ch : targets[t])
        if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T' && ch != 'X' && nodd++ == 0) first_t = t;
    if (nodd) {
      fprintf(stderr, "Warning: %zu target bytes are none of ACGTX (first in target %zu); they are treated as X\n", nodd, first_t);
      log.printf("%zu target bytes are none of ACGTX (first in target %zu): treated as X; the reference compares them "
                 "literally, so they would mismatch an X of a read there", nodd, first_t);
    }
  }
  std::map<uint64_t, std::string> id_rest;
  for (auto& l : split_lines(read_maybe_sz(cfg.GeneIdFileName))) {
    size_t t = l.find('\t');
    if (t == std::string::npos) continue;
    id_rest.emplace(strtoull(l.substr(0, t).c_str(), nullptr, 10), l.substr(t + 1));
  }

  clk.lap("windows check, target files");
  fputs("Screening...\nConfirming...\n", stderr);
  musc_stats st;
  memset(&st, 0, sizeof st);
  std::vector<musc_hit> hits = run_hot_path(cfg, reads, targets, log, &st);
  if (st.n_overflow_blocks)
    fprintf(stderr, "Warning: %llu window-key blocks may exceed MaxMatches; results keep all their matches\n",
            (unsigned long long)st.n_overflow_blocks);

  clk.lap("hot path (init, load, match)");
  fputs("Combining windows...\nJoining gene names...\nJoining read names...\n", stderr);
  const std::string res = results_text(hits.data(), hits.size(), reads, targets, id_rest);
  spit(cfg.ResultsFileName, res);
  clk.lap("results.txt");

  // the three side outputs only read `res`: one thread each
  fputs("Writing non-matching sequences...\nGenerating read statistics...\nGenerating gene statistics...\n", stderr);
  {
    std::string err_side[3];
    auto guarded = [&](int i, auto fn) {
      return std::thread([&, i, fn] {
        try { fn(); } catch (const std::exception& e) { err_side[i] = e.what(); } catch (...) { err_side[i] = "failed"; }
      });
    };
    std::thread t0 = guarded(0, [&] { spit(nonmatch_name(cfg.ResultsFileName), nonmatch_text(res, reads)); });
    std::thread t1 = guarded(1, [&] { spit(stats_name(cfg.ResultsFileName, "_readstats"), readstats_text(res)); });
    std::thread t2 = guarded(2, [&] { spit(stats_name(cfg.ResultsFileName, "_genestats"), genestats_text(res)); });
    t0.join(); t1.join(); t2.join();
    for (auto& e : err_side) if (!e.empty()) throw Die(1, "side output: " + e);
  }
  clk.lap("nonmatch + stats files");
  return 0;
}

}  // namespace musc
