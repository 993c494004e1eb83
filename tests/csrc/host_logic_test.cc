// Sanitizer harness for e2e_tts_amd/csrc/host_logic.h (the engine's HIP-free host logic: blob header / directory validation, config
// validation, vocoder halo, conv_gemm tile choice).  Built by tests/test_host_sanitizer.py with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all
// and fed valid, truncated and corrupt inputs; any out-of-bounds read or integer overflow aborts the process (non-zero exit).
//   host_logic_test blob <file>         -> "OK <n_tensors> <sum of numel>" | "ERR <message>"
//   host_logic_test config <file>       -> "OK halo=<frames> stage_rows=<r0,r1,..>" | "ERR <message>"     (file = raw e2etts_config bytes)
//   host_logic_test tiles B T Cout      -> "few=<0|1> half=<0|1>"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../e2e_tts_amd/csrc/host_logic.h"

using namespace e2etts;

static std::vector<char> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc >= 3 && !strcmp(argv[1], "blob")) {
    // the engine's sequence (engine.hip: e2etts_load_weights / bind_resident_blob) on a host copy of exactly `nbytes` bytes:
    // nothing beyond the buffer may be read, whatever the header claims
    const std::vector<char> buf = slurp(argv[2]);
    const size_t nbytes = buf.size();
    if (nbytes < sizeof(BlobHeader)) { puts("ERR weight blob too small"); return 0; }
    BlobHeader h;
    memcpy(&h, buf.data(), sizeof h);
    if (const char* m = blob_check_header(h, nbytes)) { printf("ERR %s\n", m); return 0; }
    std::vector<BlobEntry> dir(h.n_entries);
    if (h.n_entries) memcpy(dir.data(), buf.data() + sizeof h, (size_t)h.n_entries * sizeof(BlobEntry));  // in bounds: dir_end <= data_offset <= nbytes
    std::vector<BlobTensor> ts;
    std::string bad;
    if (const char* m = blob_check_directory(h, dir.data(), nbytes, ts, bad)) { printf("ERR tensor '%s': %s\n", bad.c_str(), m); return 0; }
    unsigned long long total = 0;
    double checksum = 0;
    for (const BlobTensor& t : ts) {
      total += t.numel;
      if (t.numel) {  // touch the first and the last element the engine would hand to a kernel
        float a, b;
        memcpy(&a, buf.data() + t.offset, 4);
        memcpy(&b, buf.data() + t.offset + (t.numel - 1) * 4, 4);
        checksum += a + b;
      }
    }
    printf("OK %zu %llu\n", ts.size(), total);
    return checksum == 12345.678 ? 1 : 0;
  }
  if (argc >= 3 && !strcmp(argv[1], "config")) {
    const std::vector<char> buf = slurp(argv[2]);
    if (buf.size() != sizeof(e2etts_config)) { printf("ERR config file has %zu bytes, struct has %zu\n", buf.size(), sizeof(e2etts_config)); return 0; }
    e2etts_config c;
    memcpy(&c, buf.data(), sizeof c);
    if (const char* m = config_check(c)) { printf("ERR %s\n", m); return 0; }
    long long h[E2ETTS_MAX_STAGES + 1];
    vocoder_stage_halo_rows(c, h);
    printf("OK halo=%d stage_rows=", vocoder_halo_frames(c));
    for (int i = 0; i <= c.voc_stages; ++i) printf(i ? ",%lld" : "%lld", h[i]);
    printf("\n");
    return 0;
  }
  if (argc >= 5 && !strcmp(argv[1], "tiles")) {
    const int B = atoi(argv[2]), T = atoi(argv[3]), Cout = atoi(argv[4]);
    printf("few=%d half=%d\n", tile_few_rows(B, T, Cout) ? 1 : 0, tile_half_rows(B, T, Cout) ? 1 : 0);
    return 0;
  }
  if (argc >= 5 && !strcmp(argv[1], "ragged")) {  // ragged T Cout rows_0 rows_1 ...: the counts a ragged launch is sized on, and its tile choice
    const int T = atoi(argv[2]), Cout = atoi(argv[3]);
    std::vector<int32_t> rows;
    for (int i = 4; i < argc; ++i) rows.push_back((int32_t)atoll(argv[i]));
    long long t128 = 0, n = 0;
    ragged_counts(rows.data(), (int)rows.size(), T, Cout, &t128, &n);
    printf("t128=%lld rows=%lld few=%d half=%d\n", t128, n, tile_few_rows_n(t128, n, Cout) ? 1 : 0, tile_half_rows_n(t128, n, Cout) ? 1 : 0);
    return 0;
  }
  fprintf(stderr, "usage: host_logic_test blob|config <file> | tiles B T Cout | ragged T Cout rows...\n");
  return 2;
}
