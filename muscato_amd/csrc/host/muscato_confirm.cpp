// `muscato_confirm config.json k [tmpdir]` -- file-level drop-in for
// cmd/muscato_confirm/main.go: reads TempDir/win_k_sorted.txt.sz ("key \t left \t right",
// the reads that take part in window k) and GeneFileName, writes TempDir/rmatch_k.txt.sz
// (lines "read \t targetsub \t pos \t nmiss \t %011d", :221-230), log in
// LogDir/muscato_confirm_<k>.log.  smatch_k.txt.sz is not needed: the GPU index finds the same
// key matches the sorted candidate file would supply (every true k-mer match is in smatch_k,
// SURVEY.md 8a note H).  MaxMatches / MatchMode are honoured literally (apply_maxmatches).
#include "muscato_host.hpp"

int main(int argc, char** argv) {
  using namespace musc;
  try {
    if (argc != 3 && argc != 4) {
      fprintf(stderr, "%s: wrong number of arguments", argv[0]);
      return 1;
    }
    Config cfg;
    config_from_json(slurp(argv[1]), cfg);
    const std::string tmpdir = cfg.TempDir.empty() ? (argc == 4 ? argv[3] : ".") : cfg.TempDir;
    char* end = nullptr;
    const long win = strtol(argv[2], &end, 10);
    if (*end || win < 0 || win >= (long)cfg.Windows.size()) throw Die(1, "muscato_confirm: bad window index");
    Logger log;
    log.open(join_path(cfg.LogDir.empty() ? "." : cfg.LogDir, "muscato_confirm_" + std::to_string(win) + ".log"));
    if (cfg.MaxMatches == 0) cfg.MaxMatches = 1000000;
    if (cfg.PMatch == 0) cfg.PMatch = 1;
    if (cfg.MatchMode.empty()) cfg.MatchMode = "best";

    std::vector<UniqueRead> reads;
    for (auto& l : split_lines(read_maybe_sz(join_path(tmpdir, "win_" + std::to_string(win) + "_sorted.txt.sz")))) {
      const size_t t1 = l.find('\t'), t2 = t1 == std::string::npos ? t1 : l.find('\t', t1 + 1);
      if (t2 == std::string::npos) continue;
      reads.push_back(UniqueRead{l.substr(t1 + 1, t2 - t1 - 1) + l.substr(0, t1) + l.substr(t2 + 1), 1, ""});
    }
    std::vector<std::string> targets;
    for (auto& l : split_lines(read_maybe_sz(cfg.GeneFileName))) targets.push_back(l.substr(0, l.find('\t')));
    log.printf("sourcefile: %zu reads; %zu targets", reads.size(), targets.size());

    Config one = cfg;
    one.Windows = {cfg.Windows[win]};
    one.MinDinuc = 0;  // muscato_window_reads already applied the entropy gate
    one.GPUs = 1;
    std::string out;
    if (!reads.empty()) {
      musc_ctx* c = nullptr;
      if (musc_init(cfg.Device, &c)) throw Die(1, std::string("muscato_confirm: ") + musc_last_error(nullptr));
      const Concat db = concat(targets.begin(), targets.end(), [](const std::string& s) -> const std::string& { return s; });
      const Concat rd = concat(reads.begin(), reads.end(), [](const UniqueRead& u) -> const std::string& { return u.seq; });
      musc_params P = to_params(one);
      P.apply_mmtol = 0;
      musc_hit* h = nullptr;
      uint64_t n = 0;
      if (musc_db_load_ascii(c, db.buf.data(), db.off.data(), (uint32_t)targets.size(), 0) ||
          musc_reads_load_ascii(c, rd.buf.data(), rd.off.data(), reads.size(), 0) || musc_match(c, &P, &h, &n))
        throw Die(1, std::string("muscato_confirm: ") + musc_last_error(c));
      std::vector<musc_hit> hits(h, h + n);
      musc_free_hits(h);
      musc_stats st;
      musc_get_stats(c, &st);
      if (st.n_overflow_blocks != 0 && st.n_overflow_blocks != ~0ull) {
        uint32_t *pr = nullptr, *pw = nullptr;
        uint64_t np = 0;
        if (musc_overflow_probes(c, &pr, &pw, &np)) throw Die(1, std::string("muscato_confirm: ") + musc_last_error(c));
        size_t ntrunc = 0;
        hits = apply_maxmatches(one, reads, targets, std::move(hits), pr, pw, np, &ntrunc);
        musc_free_u32(pr);
        musc_free_u32(pw);
        log.printf("%zu blocks truncated at MaxMatches", ntrunc);
      }
      musc_destroy(c);
      for (auto& x : hits) {
        const std::string& r = reads[x.read_idx].seq;
        out += r;
        out += '\t';
        out.append(targets[x.gene_idx], x.pos, r.size());
        char b[64];
        snprintf(b, sizeof b, "\t%u\t%u\t%011u\n", x.pos, x.nmiss, x.gene_idx);
        out += b;
      }
      log.printf("%zu matches", hits.size());
    } else {
      log.printf("No matches found, done.");
    }
    spit(join_path(tmpdir, "rmatch_" + std::to_string(win) + ".txt.sz"), sz_encode(out));
    log.printf("done");
    return 0;
  } catch (const Die& d) {
    fprintf(stderr, "%s\n", d.what());
    return d.code ? d.code : 1;
  } catch (const std::exception& e) {
    fprintf(stderr, "muscato_confirm: %s\n", e.what());
    return 2;
  }
}
