// tools/asan/host_fuzz.cpp — driver of the sanitizer build (make asan): runs the product's HOST path (proto3 wire decode, typed tensor
// accessors, fp16 codec, converter loop, planner, BN fold + fp16 cast + weight packing) over model files and over structured
// mutations of them, under AddressSanitizer + UndefinedBehaviorSanitizer.  CPU only: the kernel entry points are stubs.
//   host_fuzz [--iters N] [--seed S] model.onnx ...
// Exit code 0 = no sanitizer report (error CODES from the library are expected and fine); any report aborts the process.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "smelter_hip.h"

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return g_rng; }

static int g_ok = 0, g_err = 0;
static void exercise(const std::vector<uint8_t>& bytes) {
    smelter_config cfg; smelter_config_default(&cfg);
    smelter_graph* g = nullptr;
    int rc = smelter_graph_create(bytes.data(), bytes.size(), &cfg, &g);
    if (rc) { g_err++; (void)smelter_last_error_string(); return; }
    smelter_shape shapes[8]; (void)smelter_graph_output_shapes(g, shapes, 8);
    (void)smelter_graph_format(g); (void)smelter_graph_num_nodes(g);
    size_t nb = 0;
    rc = smelter_graph_weight_bytes(g, 2, &nb);
    if (!rc && nb <= ((size_t)1 << 30)) {
        std::vector<uint8_t> arena(nb);
        rc = smelter_graph_pack_weights(g, 2, arena.data(), arena.size());
    }
    if (rc) g_err++; else g_ok++;
    smelter_graph_destroy(g);
}

int main(int argc, char** argv) {
    int iters = 2000; std::vector<std::string> files;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--seed") && i + 1 < argc) g_rng ^= strtoull(argv[++i], nullptr, 0) * 0x2545F4914F6CDD1Dull;
        else files.push_back(argv[i]);
    }
    std::vector<std::vector<uint8_t>> corpus;
    for (auto& f : files) {
        FILE* fp = fopen(f.c_str(), "rb"); if (!fp) { fprintf(stderr, "cannot open %s\n", f.c_str()); return 2; }
        std::vector<uint8_t> b; uint8_t buf[65536]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, fp)) > 0) b.insert(b.end(), buf, buf + n);
        fclose(fp); corpus.push_back(std::move(b));
    }
    for (auto& b : corpus) exercise(b);                       // the files as they are (valid models and hand-made error cases)
    const int clean_ok = g_ok, clean_err = g_err;
    // half <-> float codec and the weight re-layout on their own
    { std::vector<uint16_t> h(65536); std::vector<float> f(65536); for (int i = 0; i < 65536; i++) h[i] = (uint16_t)i;
      sm_half_to_float(h.data(), f.data(), h.size()); sm_float_to_half(f.data(), h.data(), f.size());
      std::vector<float> w(5 * 3 * 2 * 4), o(w.size()); for (size_t i = 0; i < w.size(); i++) w[i] = (float)i;
      (void)sm_reformat_conv_weight(w.data(), o.data(), 4, 5, 3, 2, 4, 0); (void)sm_reformat_conv_weight(w.data(), o.data(), 4, 3, 5, 2, 4, 1); }
    for (int it = 0; it < iters && !corpus.empty(); it++) {
        std::vector<uint8_t> m = corpus[rnd() % corpus.size()];
        if (m.size() > (1u << 20)) continue;                  // mutate the small models; the big ones only run clean
        const int kind = (int)(rnd() % 6), reps = 1 + (int)(rnd() % 4);
        for (int r = 0; r < reps && !m.empty(); r++) {
            const size_t pos = rnd() % m.size();
            switch (kind) {
            case 0: m.resize(pos); break;                                                   // truncation
            case 1: m[pos] ^= (uint8_t)(1u << (rnd() % 8)); break;                            // bit flip
            case 2: m[pos] = (uint8_t)rnd(); break;                                         // byte overwrite
            case 3: { const uint8_t big[10] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0x01};   // a maximal varint (length / dim / count)
                      for (size_t k = 0; k < sizeof big && pos + k < m.size(); k++) m[pos + k] = big[k]; break; }
            case 4: { const size_t n = 1 + rnd() % 64; m.insert(m.begin() + pos, n, (uint8_t)rnd()); break; }   // junk insertion
            case 5: { const size_t a = rnd() % m.size(), n = 1 + rnd() % 32;                                    // splice a chunk from elsewhere
                      for (size_t k = 0; k < n && a + k < m.size() && pos + k < m.size(); k++) m[pos + k] = m[a + k]; break; }
            }
        }
        exercise(m);
    }
    printf("host_fuzz: %zu files (%d ok, %d error codes), %d mutations -> %d ok / %d error codes, no sanitizer report\n",
           corpus.size(), clean_ok, clean_err, iters, g_ok - clean_ok, g_err - clean_err);
    return 0;
}
