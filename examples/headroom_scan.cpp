// headroom_scan -- a C++ host on top of the C ABI that does what headroom's scriptable analyse step does for
// WAV / AIFF files (src/cli.rs:165-227 with --analyze-only, and --apply for the lossless gain of
// src/processor.rs:53-87): load the files, measure them in ONE hr_analyze_batch, apply the gain decision
// (src/analyzer.rs:304-326) and print the CSV report of src/report.rs:29-59.  It is what a maintainer's Rust code
// would do through the same entry points; Rust is not in this image, so the host of record is C++.
//
//   headroom_scan [--tp-target dBTP] [--apply] file.wav file.aif ...
//
// Files that cannot be read or are not PCM are reported on stderr and skipped, like the warnings of
// src/cli.rs:419-430; the exit status is 1 only when nothing could be analysed or the device is missing.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/headroom_hip.h"

int main(int argc, char **argv)
{
    double tp_target = -0.5;  // TpTargetMode::Uniform(TARGET_TRUE_PEAK_HIGH_BITRATE), src/analyzer.rs:16-26: all inputs here are lossless
    bool apply = false;
    std::vector<std::string> paths;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--tp-target") && i + 1 < argc) tp_target = std::atof(argv[++i]);
        else if (!std::strcmp(argv[i], "--apply")) apply = true;
        else paths.push_back(argv[i]);
    }
    if (paths.empty()) {
        std::fprintf(stderr, "usage: headroom_scan [--tp-target dBTP] [--apply] file.wav file.aif ...\n");
        return 2;
    }
    if (hr_device_count() <= 0) {  // check_ffmpeg's analogue, src/cli.rs:30
        std::fprintf(stderr, "headroom_scan: %s\n", hr_strerror(HR_ERR_NO_DEVICE));
        return 1;
    }
    hr_ctx *ctx = nullptr;
    int rc = hr_init(nullptr, &ctx);
    if (rc) { std::fprintf(stderr, "headroom_scan: hr_init: %s\n", hr_strerror(rc)); return 1; }

    struct Loaded { std::string path; void *pcm; float w[HR_MAX_CH]; };
    std::vector<Loaded> files;
    files.reserve(paths.size());  // the tracks point at the weights inside: no reallocation after this
    std::vector<hr_track> tracks;
    for (const std::string &p : paths) {
        files.push_back(Loaded{p, nullptr, {0}});
        hr_track t;
        rc = hr_pcm_load_file(p.c_str(), &files.back().pcm, &t, files.back().w);
        if (rc) {
            std::fprintf(stderr, "warning: %s: %s\n", p.c_str(), hr_strerror(rc));
            files.pop_back();
            continue;
        }
        tracks.push_back(t);
    }
    if (tracks.empty()) { hr_destroy(ctx); return 1; }
    std::vector<hr_result> res(tracks.size());
    rc = hr_analyze_batch(ctx, tracks.data(), tracks.size(), res.data());
    if (rc) { std::fprintf(stderr, "headroom_scan: hr_analyze_batch: %s\n", hr_strerror(rc)); return 1; }

    char line[4096];
    hr_report_csv_header(line, sizeof line);
    std::fputs(line, stdout);
    const hr_tp_target mode{0, tp_target, tp_target};
    int analysed = 0;
    for (size_t i = 0; i < tracks.size(); i++) {
        const std::string &p = files[i].path;
        const std::string name = p.substr(p.find_last_of('/') == std::string::npos ? 0 : p.find_last_of('/') + 1);
        hr_analysis a;
        if (res[i].status) { std::fprintf(stderr, "warning: %s: %s\n", p.c_str(), hr_strerror(res[i].status)); continue; }
        // the decision sees the measurement as the reference does: through loudnorm's "%.2f" text (src/analyzer.rs:273-280)
        if (hr_gain_decision_for_result(&res[i], HR_KIND_LOSSLESS, 0, 0, &mode, &a)) {
            // src/analyzer.rs:284-290
            std::fprintf(stderr, "warning: %s: Non-finite loudness measurement (silent or unreadable audio)\n", p.c_str());
            continue;
        }
        if (hr_report_csv_row(name.c_str(), &a, line, sizeof line) > 0) std::fputs(line, stdout);
        analysed++;
        if (apply && a.gain_method == HR_GAIN_FFMPEG_LOSSLESS) {
            uint64_t clipped = 0;
            rc = hr_pcm_gain_file(ctx, p.c_str(), a.effective_gain, &clipped);
            if (rc) std::fprintf(stderr, "warning: %s: gain not applied: %s\n", p.c_str(), hr_strerror(rc));
            else std::fprintf(stderr, "applied %+.1f dB to %s (%llu samples clipped)\n", a.effective_gain, p.c_str(),
                              static_cast<unsigned long long>(clipped));
        }
    }
    for (Loaded &f : files) hr_free_pinned(f.pcm);
    hr_destroy(ctx);
    return analysed ? 0 : 1;
}
