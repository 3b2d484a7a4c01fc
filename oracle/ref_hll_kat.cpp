// ref_hll_kat.cpp -- ORACLE tooling (test infrastructure, NOT product code).
//
// Known-answer generator: links the REFERENCE's own sketch::hll_t (vendored dnbaker/sketch, compiled
// from the headers where they lie under /root/reference -- nothing is copied) and prints, as C99 hex
// floats, report() of every .hll file named on the command line followed by union_size() of every
// pair.  Output feeds tests/golden/*.kat (see tests/golden/make_golden.py).  Built only in the
// authoring container by oracle/Makefile into oracle/_ref/; never needed on the GPU box.
//
// usage: hll_kat <suffix> file1 file2 ...     (suffix e.g. ".hll" or ".hll_8"; "" = names are full paths)
#include "sketch/sketch.h"
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s suffix files...\n", argv[0]); return 1; }
    std::string suffix = argv[1];
    std::vector<std::shared_ptr<sketch::hll_t>> v;
    for (int i = 2; i < argc; ++i) v.push_back(std::make_shared<sketch::hll_t>(std::string(argv[i]) + suffix));
    for (size_t i = 0; i < v.size(); ++i) std::printf("R %zu %a\n", i, v[i]->report());
    for (size_t i = 0; i < v.size(); ++i)
        for (size_t k = i + 1; k < v.size(); ++k)
            std::printf("U %zu %zu %a\n", i, k, v[i]->union_size(*v[k]));
    return 0;
}
