// `muscato_screen config.json [tmpdir]` -- file-level drop-in for cmd/muscato_screen/main.go:
// reads TempDir/reads_sorted.txt.sz and GeneFileName, writes TempDir/bmatch_k.txt.sz
// (lines "mseq \t left \t right \t %011d \t pos", :371-399) for every window, log in
// LogDir/muscato_screen.log.  The candidates come from the GPU index instead of the Bloom
// sketch: exactly the target positions whose k-mer equals some read's window k-mer and where
// that read fits -- a subset of what the reference writes (no Bloom false positives, no
// positions its confirm rejects for running past the target end), so the downstream sort +
// muscato_confirm (the reference's or this repo's) produce the same rmatch_k.
#include "muscato_host.hpp"

int main(int argc, char** argv) {
  using namespace musc;
  try {
    if (argc != 2 && argc != 3) {
      fprintf(stderr, "%s: wrong number of arguments", argv[0]);
      return 1;
    }
    Config cfg;
    config_from_json(slurp(argv[1]), cfg);
    const std::string tmpdir = cfg.TempDir.empty() ? (argc == 3 ? argv[2] : ".") : cfg.TempDir;
    Logger log;
    log.open(join_path(cfg.LogDir.empty() ? "." : cfg.LogDir, "muscato_screen.log"));
    std::vector<std::string> reads;
    for (auto& l : split_lines(read_maybe_sz(join_path(tmpdir, "reads_sorted.txt.sz")))) {
      auto f = fields(l);  // bytes.Fields(line)[0], cmd/muscato_screen/main.go:171-172
      if (!f.empty()) reads.push_back(f[0]);
    }
    std::vector<std::string> targets;
    for (auto& l : split_lines(read_maybe_sz(cfg.GeneFileName))) targets.push_back(l.substr(0, l.find('\t')));
    log.printf("%zu reads, %zu targets", reads.size(), targets.size());

    musc_ctx* c = nullptr;
    if (musc_init(cfg.Device, &c)) throw Die(1, std::string("muscato_screen: ") + musc_last_error(nullptr));
    const Concat db = concat(targets.begin(), targets.end(), [](const std::string& s) -> const std::string& { return s; });
    const Concat rd = concat(reads.begin(), reads.end(), [](const std::string& s) -> const std::string& { return s; });
    if (musc_db_load_ascii(c, db.buf.data(), db.off.data(), (uint32_t)targets.size(), 0) ||
        musc_reads_load_ascii(c, rd.buf.data(), rd.off.data(), reads.size(), 0))
      throw Die(1, std::string("muscato_screen: ") + musc_last_error(c));
    const int ww = cfg.WindowWidth;
    for (size_t k = 0; k < cfg.Windows.size(); k++) {
      Config one = cfg;
      one.Windows = {cfg.Windows[k]};
      one.MaxMismatch = 65535;  // every k-mer match of a fitting read is a candidate
      musc_params P = to_params(one);
      P.apply_mmtol = 0;
      P.skip_block_check = 1;
      P.n_shards = 0;
      musc_hit* h = nullptr;
      uint64_t n = 0;
      if (musc_match(c, &P, &h, &n)) throw Die(1, std::string("muscato_screen: ") + musc_last_error(c));
      const int q1 = cfg.Windows[k], q2 = q1 + ww;
      std::vector<std::pair<uint32_t, uint32_t>> cand(n);
      for (uint64_t i = 0; i < n; i++) cand[i] = std::make_pair(h[i].gene_idx, h[i].pos + (uint32_t)q1);
      musc_free_hits(h);
      std::sort(cand.begin(), cand.end());
      cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
      std::string out;
      for (auto& cd : cand) {
        const std::string& t = targets[cd.first];
        const int64_t jx = cd.second, T = (int64_t)t.size();
        out.append(t, jx, ww);
        out += '\t';
        int64_t jy, jz;
        if (jx == 0) {  // cmd/muscato_screen/main.go:303-315
          jy = ww;
          jz = std::min<int64_t>(100 - q2, T);
        } else {        // :341-361
          out.append(t, jx - q1, q1);
          jy = jx + ww;
          jz = std::min<int64_t>(jy + cfg.MaxReadLength - q2, T);
        }
        out += '\t';
        if (jz > jy) out.append(t, jy, jz - jy);
        char b[40];
        snprintf(b, sizeof b, "\t%011u\t%lld\n", cd.first, (long long)jx);
        out += b;
      }
      spit(join_path(tmpdir, "bmatch_" + std::to_string(k) + ".txt.sz"), sz_encode(out));
      log.printf("window %zu: %zu candidates", k, cand.size());
    }
    musc_destroy(c);
    log.printf("Done checking target sequences for matches");
    return 0;
  } catch (const Die& d) {
    fprintf(stderr, "%s\n", d.what());
    return d.code ? d.code : 1;
  } catch (const std::exception& e) {
    fprintf(stderr, "muscato_screen: %s\n", e.what());
    return 2;
  }
}
