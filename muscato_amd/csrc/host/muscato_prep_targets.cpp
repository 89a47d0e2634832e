// `muscato_prep_targets [-rev] genefile` -- cmd/muscato_prep_targets/main.go: FASTA or
// "id<TAB>sequence" text (plain, .gz or .sz) -> musc_<file>.sz + musc_ids_<file>.sz.
#include "muscato_host.hpp"

int main(int argc, char** argv) {
  try {
    static const std::vector<musc::FlagSpec> flags = {{"rev", 'b', "Include reverse complement sequences"}};
    std::vector<std::string> rest;
    auto fl = musc::parse_flags(argc, argv, flags, &rest);
    if (rest.size() != 1) {
      fputs("muscato_prep_targets: usage\n  muscato_prep_targets [-rev] genefile\n\n", stderr);
      return 1;
    }
    std::string seqout, idout;
    musc::prep_targets_file(rest[0], fl.count("rev") && fl["rev"] == "true", &seqout, &idout);
    fprintf(stderr, "Gene sequence file: %s\nGene ids file: %s\n", seqout.c_str(), idout.c_str());
    return 0;
  } catch (const musc::Die& d) {
    fputs(d.what(), stderr);
    return d.code;
  } catch (const std::exception& e) {
    fprintf(stderr, "muscato_prep_targets: %s\n", e.what());
    return 2;
  }
}
