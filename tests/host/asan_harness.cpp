// asan_harness.cpp -- the host-only product sources (container probe / loader, the file half of the lossless gain, the
// analyzer's scalar logic, the rate tables) under AddressSanitizer + UndefinedBehaviorSanitizer, without a GPU.
//
// Built by `make -C headroom_amd/csrc asan` from hr_ingest.cpp, hr_gainfile.cpp, hr_analyzer.cpp, hr_tables.cpp and this
// file; the three entry points those sources take from the GPU side of the library are stand-ins here (pinned memory =
// malloc, hr_apply_gain = a byte pattern: the file logic around it is what is under test).  Test infrastructure only.
//
//   asan_harness fuzz <dir with seed files> <iterations> <seed>
//       mutates the seed WAV / AIFF images (truncation, byte flips, extreme 32-bit fields, insertions) and drives
//       hr_pcm_probe on the buffer and hr_pcm_load_file / hr_pcm_gain_file on a file.  Checks: nothing reads or writes out
//       of bounds (the sanitizers), what the probe accepts lies inside the file, and the never-half-written contract of
//       src/processor.rs:58,80-86 -- a rejected file is byte-identical afterwards and leaves no temp file behind, an
//       accepted one probes again as 24-bit PCM of the same geometry.
//   asan_harness logic
//       the gain decision, the two-decimal text round trip and the CSV row on awkward numbers (NaN, infinities, huge gains,
//       names that need quoting, buffers one byte short), and the rate tables of the usual and of awkward rates.
#include <sys/stat.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/headroom_hip.h"
#include "../../headroom_amd/csrc/hr_tables.hpp"

// ---- stand-ins for the GPU side ----
extern "C" void *hr_alloc_pinned(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
extern "C" void hr_free_pinned(void *p) { std::free(p); }
extern "C" int hr_apply_gain(hr_ctx *, const hr_track *track, double, int, void *out, uint64_t *clipped)
{
    const uint64_t n = track->frames * track->channels * 3;
    unsigned char *o = static_cast<unsigned char *>(out);
    for (uint64_t i = 0; i < n; i++) o[i] = static_cast<unsigned char>(i * 7 + 1);
    if (clipped) *clipped = 0;
    return HR_OK;
}

namespace {

uint64_t rng_state = 88172645463325252ull;
uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
uint64_t below(uint64_t n) { return n ? rnd() % n : 0; }

bool read_all(const std::string &path, std::vector<unsigned char> &out)
{
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? n : 0);
    const size_t got = out.empty() ? 0 : std::fread(out.data(), 1, out.size(), f);
    std::fclose(f);
    return got == out.size();
}

bool write_all(const std::string &path, const std::vector<unsigned char> &b)
{
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const size_t wr = b.empty() ? 0 : std::fwrite(b.data(), 1, b.size(), f);
    return std::fclose(f) == 0 && wr == b.size();
}

bool exists(const std::string &path)
{
    struct stat st;
    return stat(path.c_str(), &st) == 0;
}

int bytes_of(int fmt) { return fmt == HR_FMT_S16 ? 2 : fmt == HR_FMT_S24 ? 3 : fmt == HR_FMT_F64 ? 8 : 4; }

#define CHECK(cond, ...)                                                      \
    do {                                                                      \
        if (!(cond)) {                                                        \
            std::fprintf(stderr, "asan_harness: %s failed: ", #cond);         \
            std::fprintf(stderr, __VA_ARGS__);                                \
            std::fprintf(stderr, "\n");                                       \
            std::exit(2);                                                     \
        }                                                                     \
    } while (0)

int fuzz(const char *dir, long iterations, uint64_t seed)
{
    rng_state ^= seed * 0x9E3779B97F4A7C15ull + 1;
    std::vector<std::vector<unsigned char>> seeds;
    std::vector<std::string> exts;
    for (int i = 0; i < 64; i++)
        for (const char *ext : {"wav", "aiff"}) {
            std::vector<unsigned char> b;
            if (read_all(std::string(dir) + "/seed" + std::to_string(i) + "." + ext, b)) { seeds.push_back(b); exts.push_back(ext); }
        }
    CHECK(!seeds.empty(), "no seed files in %s", dir);
    long probed_ok = 0, loaded_ok = 0, gained_ok = 0;
    hr_ctx *fake_ctx = reinterpret_cast<hr_ctx *>(&rng_state);  // (never dereferenced: hr_apply_gain is the stand-in above)
    for (long it = 0; it < iterations; it++) {
        const size_t si = below(seeds.size());
        std::vector<unsigned char> b = seeds[si];
        switch (below(5)) {
        case 0: b.resize(below(b.size() + 1)); break;
        case 1: for (int k = 1 + static_cast<int>(below(5)); k > 0 && !b.empty(); k--) b[below(b.size())] = static_cast<unsigned char>(rnd()); break;
        case 2: if (b.size() > 4) { const size_t at = below(b.size() - 4); static const unsigned char ex[4] = {0, 0xFF, 0x7F, 0x80}; for (int k = 0; k < 4; k++) b[at + k] = ex[below(4)]; } break;
        case 3: { const size_t at = below(b.size() + 1); std::vector<unsigned char> ins(1 + below(8)); for (auto &x : ins) x = static_cast<unsigned char>(rnd()); b.insert(b.begin() + at, ins.begin(), ins.end()); } break;
        default: break;  // the seed as it is
        }
        hr_pcm_info info;
        // an exact-size heap copy: a read one byte past the image is an error the sanitizer sees
        unsigned char *exact = static_cast<unsigned char *>(std::malloc(b.size() ? b.size() : 1));
        if (!b.empty()) std::memcpy(exact, b.data(), b.size());
        const int prc = hr_pcm_probe(exact, b.size(), &info);
        std::free(exact);
        if (prc == HR_OK) {
            probed_ok++;
            CHECK(info.channels >= 1 && info.channels <= HR_MAX_CH && info.sample_rate > 0, "geometry %u ch %u Hz", info.channels, info.sample_rate);
            CHECK(info.data_offset + info.frames * info.channels * bytes_of(info.fmt) <= b.size(), "samples past the end of the image");
        }
        const std::string path = std::string(dir) + "/case." + exts[si];
        const std::string tmp = std::string(dir) + "/case.tmp." + exts[si];
        CHECK(write_all(path, b), "cannot write %s", path.c_str());
        void *pcm = nullptr;
        hr_track track;
        float w[HR_MAX_CH];
        const int lrc = hr_pcm_load_file(path.c_str(), &pcm, &track, w);
        if (lrc == HR_OK) {
            loaded_ok++;
            CHECK(pcm && track.pcm == pcm && track.channels >= 1 && track.channels <= HR_MAX_CH, "loaded track");
            // every byte the loader says it holds can be read
            const unsigned char *p = static_cast<const unsigned char *>(pcm);
            unsigned acc = 0;
            for (uint64_t i = 0; i < track.frames * track.channels * bytes_of(track.fmt); i++) acc += p[i];
            (void)acc;
            hr_free_pinned(pcm);
        } else {
            CHECK(pcm == nullptr, "a failed load left a buffer behind");
        }
        uint64_t clipped = 0;
        const int grc = hr_pcm_gain_file(fake_ctx, path.c_str(), -3.0, &clipped);
        std::vector<unsigned char> after;
        CHECK(read_all(path, after), "the file is gone after hr_pcm_gain_file (rc %d)", grc);
        CHECK(!exists(tmp), "temp file left behind (rc %d)", grc);
        if (grc != HR_OK) {
            CHECK(after == b, "a rejected file (rc %d) was modified", grc);
        } else {
            gained_ok++;
            hr_pcm_info again;
            CHECK(hr_pcm_probe(after.data(), after.size(), &again) == HR_OK, "the rewritten file does not probe");
            CHECK(again.fmt == HR_FMT_S24 && again.channels == track.channels && again.frames == track.frames && again.sample_rate == track.sample_rate,
                  "the rewritten file's geometry: %u ch %llu frames", again.channels, static_cast<unsigned long long>(again.frames));
        }
    }
    std::remove((std::string(dir) + "/case.wav").c_str());
    std::remove((std::string(dir) + "/case.aiff").c_str());
    std::printf("asan_harness fuzz: %ld mutated files: probe ok %ld, load ok %ld, gain ok %ld; no sanitizer report, every rejected file intact\n",
                iterations, probed_ok, loaded_ok, gained_ok);
    return 0;
}

int logic()
{
    const double inf = std::numeric_limits<double>::infinity(), nan = std::numeric_limits<double>::quiet_NaN();
    const double tps[] = {-inf, -144.0, -3.0, -1.5001, -1.05, -0.5451, -0.05, 0.0, 0.049, 2.55, 100.0, 1e308, inf, nan};
    const double lufs[] = {-inf, -70.0, -23.0, -6.83, 0.0, 1e300, nan};
    hr_tp_target split{1, -0.5, -1.0}, uni{0, -1.0, 0.0};
    int decided = 0;
    for (double tp : tps)
        for (double l : lufs)
            for (int kind = 0; kind < 3; kind++)
                for (int hb = 0; hb < 2; hb++)
                    for (const hr_tp_target *mode : {static_cast<const hr_tp_target *>(nullptr), static_cast<const hr_tp_target *>(&split), static_cast<const hr_tp_target *>(&uni)}) {
                        hr_analysis a;
                        const int rc = hr_gain_decision(l, tp, kind, hb, hb ? 4000000000u : 0u, mode, &a);
                        CHECK(rc == HR_OK || rc == HR_ERR_INVALID, "hr_gain_decision rc %d", rc);
                        hr_result r{};
                        r.integrated_lufs = l;
                        r.true_peak_dbtp = tp;
                        double ti = 0, tt = 0;
                        CHECK(hr_loudnorm_text_values(&r, &ti, &tt) == HR_OK, "text values");
                        hr_analysis b;
                        (void)hr_gain_decision_for_result(&r, kind, hb, 128, mode, &b);
                        if (rc == HR_OK) {
                            decided++;
                            for (size_t cap : {size_t{0}, size_t{1}, size_t{40}, size_t{4096}}) {
                                std::vector<char> buf(cap ? cap : 1);
                                const int n = hr_report_csv_row("a \"quoted\", name\n.wav", &a, cap ? buf.data() : nullptr, cap);
                                CHECK(n == HR_ERR_TOO_LARGE || n == HR_ERR_INVALID || (n >= 0 && static_cast<size_t>(n) < cap), "csv row length %d in %zu", n, cap);
                            }
                        }
                    }
    char hdr[8];
    CHECK(hr_report_csv_header(hdr, sizeof hdr) == HR_ERR_TOO_LARGE, "a short header buffer");
    for (uint32_t fs : {8000u, 11025u, 12345u, 22050u, 44100u, 48000u, 88200u, 96000u, 176400u, 192000u, 384000u, 768000u, 9u, 10u}) {
        const hr::RateTables *rt = hr::rate_tables(fs);
        CHECK((rt != nullptr) == (fs >= 10), "rate tables of %u Hz", fs);
        if (rt) CHECK(rt->hop == fs / 10 && rt->g.size() == static_cast<size_t>(rt->hop) * 4, "tables of %u Hz", fs);
    }
    float w[HR_MAX_CH];
    for (int ch = 1; ch <= HR_MAX_CH; ch++) {
        hr::default_weights(ch, w);
        hr::weights_from_mask(0xFFFFFFFFu, ch, w);
        hr::weights_from_mask(0x3F, ch, w);
    }
    (void)hr::ebur128_bin_edges();
    (void)hr::loudnorm_bin_bounds();
    std::printf("asan_harness logic: %d decisions, csv rows, rate tables: no sanitizer report\n", decided);
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc >= 5 && !std::strcmp(argv[1], "fuzz")) return fuzz(argv[2], std::atol(argv[3]), std::strtoull(argv[4], nullptr, 10));
    if (argc >= 2 && !std::strcmp(argv[1], "logic")) return logic();
    std::fprintf(stderr, "usage: asan_harness fuzz <seed dir> <iterations> <seed> | asan_harness logic\n");
    return 64;
}
