// `muscato` -- drop-in for the reference driver (cmd/muscato/main.go) on MI355X: same flags,
// JSON config, temp/log directory layout and output files; the screen -> sort -> confirm ->
// combine stages run on the GPU through libmuscato_hip.so.
#include "muscato_host.hpp"

int main(int argc, char** argv) {
  try {
    musc::Config cfg = musc::handle_args(argc, argv);
    musc::check_args(cfg);
    setenv("LC_ALL", "C", 1);  // setupEnvs (cmd/muscato/main.go:906-912)
    return musc::run_muscato(cfg);
  } catch (const musc::Die& d) {
    fputs(d.what(), d.code == 0 ? stdout : stderr);
    if (d.code) fputc('\n', stderr);
    return d.code;
  } catch (const std::exception& e) {
    // the reference panics (exit status 2) on any stage failure (cmd/muscato/main.go:313-315)
    fprintf(stderr, "muscato: %s\n", e.what());
    return 2;
  }
}
