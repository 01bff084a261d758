// hr_analyzer.cpp -- host-side mirror of the analyzer's scalar logic.
//
// Follows src/analyzer.rs:16-33 (constants), :157-187 (TpTargetMode::target_for) and :284-326 (finite
// check and gain decision).  GAIN_STEP is mp3rgain::GAIN_STEP_DB (src/analyzer.rs:29; mp3rgain 2.9.1,
// Cargo.lock:961-964; README.md:282 states the value: 1.5 dB, fixed by the MP3/AAC global_gain step).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../include/headroom_hip.h"

// No exception crosses the C ABI: every int-returning entry point is a function-try-block (std::bad_alloc from a host
// container -> HR_ERR_OOM, anything else -> HR_ERR_INVALID).
#ifndef HR_CATCH_ALL
#define HR_CATCH_ALL catch (const std::bad_alloc &) { return HR_ERR_OOM; } catch (...) { return HR_ERR_INVALID; }
#endif

namespace {
constexpr unsigned kHighBitrateThreshold = 256;  // analyzer.rs:26
constexpr double kGainStep = 1.5;                // analyzer.rs:29
constexpr double kMinEffectiveGain = 0.05;       // analyzer.rs:33

double target_for(const hr_tp_target &m, bool is_lossy, bool has_bitrate, unsigned kbps)
{
    if (!m.split) return m.high;              // Uniform(t)
    if (!is_lossy) return m.high;             // SplitBitrate: lossless -> high
    if (has_bitrate && kbps >= kHighBitrateThreshold) return m.high;
    return m.low;
}
}  // namespace

extern "C" int hr_gain_decision(double input_i, double input_tp, int kind, int has_bitrate, uint32_t bitrate_kbps,
                                const hr_tp_target *mode, hr_analysis *out)
try {
    if (!out || kind < HR_KIND_LOSSLESS || kind > HR_KIND_AAC) return HR_ERR_INVALID;
    hr_tp_target m{0, -0.5, -1.0};  // DEFAULT_TARGET_TRUE_PEAK, analyzer.rs:16
    if (mode) m = *mode;
    *out = hr_analysis{};
    out->input_i = input_i;
    out->input_tp = input_tp;
    // analyzer.rs:284-290: loudnorm reports "-inf" for silent audio; reject non-finite values
    if (!std::isfinite(input_i) || !std::isfinite(input_tp)) {
        out->status = HR_ERR_INVALID;
        out->gain_method = HR_GAIN_NONE;
        return HR_ERR_INVALID;
    }
    const bool is_mp3 = kind == HR_KIND_MP3, is_aac = kind == HR_KIND_AAC;
    const bool is_lossy = is_mp3 || is_aac;
    // analyzer.rs:298-302: bitrate only matters (and is only looked up) for lossy files
    out->has_bitrate = is_lossy && has_bitrate;
    out->bitrate_kbps = out->has_bitrate ? bitrate_kbps : 0;
    out->target_tp = target_for(m, is_lossy, out->has_bitrate, out->bitrate_kbps);
    out->headroom = out->target_tp - input_tp;
    if (out->headroom < kMinEffectiveGain) {
        out->gain_method = HR_GAIN_NONE;
        out->effective_gain = 0.0;
        out->lossless_gain_steps = 0;
    } else if (!is_lossy) {
        out->gain_method = HR_GAIN_FFMPEG_LOSSLESS;
        out->effective_gain = out->headroom;
        out->lossless_gain_steps = 0;
    } else {
        const int steps = static_cast<int>(std::floor(out->headroom / kGainStep));
        if (steps >= 1) {
            out->gain_method = is_aac ? HR_GAIN_AAC_LOSSLESS : HR_GAIN_MP3_LOSSLESS;
            out->effective_gain = static_cast<double>(steps) * kGainStep;
            out->lossless_gain_steps = steps;
        } else {
            out->gain_method = is_aac ? HR_GAIN_AAC_REENCODE : HR_GAIN_MP3_REENCODE;
            out->effective_gain = out->headroom;
            out->lossless_gain_steps = 0;
        }
    }
    out->status = HR_OK;
    return HR_OK;
} HR_CATCH_ALL

// What the reference actually holds after src/analyzer.rs:273-280: af_loudnorm prints `input_i` / `input_tp` with
// "%.2f" (the fixture at src/analyzer.rs:370-371 shows the format) and the analyzer parses that text with
// str::parse::<f64>, so every later step -- the finite check, headroom, the 0.05 dB and 1.5 dB thresholds, the
// one-decimal CSV fields -- sees the measurement rounded to two decimals.  printf rounds the exact binary value to
// nearest (ties to even on the decimal expansion) and strtod, like Rust's parser, returns the double nearest to the
// text; "-inf" / "inf" / "nan" survive the round trip as such.
static double through_text_2dp(double v)
{
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.2f", v);
    return std::strtod(buf, nullptr);
}

extern "C" int hr_loudnorm_text_values(const hr_result *r, double *input_i, double *input_tp)
try {
    if (!r) return HR_ERR_INVALID;
    if (input_i) *input_i = through_text_2dp(r->integrated_lufs);
    if (input_tp) *input_tp = through_text_2dp(r->true_peak_dbtp);
    return HR_OK;
} HR_CATCH_ALL

extern "C" int hr_gain_decision_for_result(const hr_result *r, int kind, int has_bitrate, uint32_t bitrate_kbps,
                                           const hr_tp_target *mode, hr_analysis *out)
try {
    if (!r || !out) return HR_ERR_INVALID;
    if (r->status != HR_OK) {  // the track failed to measure: the Err arm of analyze_file_with_target
        *out = hr_analysis{};
        out->gain_method = HR_GAIN_NONE;
        out->status = r->status;
        return r->status;
    }
    double i = 0, tp = 0;
    hr_loudnorm_text_values(r, &i, &tp);
    return hr_gain_decision(i, tp, kind, has_bitrate, bitrate_kbps, mode, out);
} HR_CATCH_ALL

// ---- report rows: src/report.rs:29-59 (CSV header and record), labels src/analyzer.rs:80-100 ---------------------
namespace {
const char *format_label(int m)  // GainMethod::format_label
{
    switch (m) {
    case HR_GAIN_MP3_LOSSLESS: case HR_GAIN_MP3_REENCODE: return "MP3";
    case HR_GAIN_AAC_LOSSLESS: case HR_GAIN_AAC_REENCODE: return "AAC";
    case HR_GAIN_FFMPEG_LOSSLESS: return "Lossless";
    default: return "-";
    }
}
const char *method_label(int m)  // GainMethod::method_label
{
    switch (m) {
    case HR_GAIN_FFMPEG_LOSSLESS: return "ffmpeg";
    case HR_GAIN_MP3_LOSSLESS: case HR_GAIN_AAC_LOSSLESS: return "native";
    case HR_GAIN_MP3_REENCODE: case HR_GAIN_AAC_REENCODE: return "re-encode";
    default: return "none";
    }
}
// the csv crate's default quoting (QuoteStyle::Necessary): quote a field that holds the delimiter, a quote or a
// line break, doubling the quotes inside
void csv_field(std::string &o, const char *f)
{
    if (!std::strpbrk(f, ",\"\r\n")) { o += f; return; }
    o += '"';
    for (const char *p = f; *p; p++) { if (*p == '"') o += '"'; o += *p; }
    o += '"';
}
int emit(const std::string &s, char *buf, size_t cap)
{
    if (!buf || cap < s.size() + 1) return HR_ERR_TOO_LARGE;
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return static_cast<int>(s.size());
}
}  // namespace

extern "C" int hr_report_csv_header(char *buf, size_t cap)
try {
    return emit("Filename,Format,Bitrate (kbps),LUFS,True Peak (dBTP),Target (dBTP),Headroom (dB),Method,Effective Gain (dB)\n", buf, cap);
} HR_CATCH_ALL

extern "C" int hr_report_csv_row(const char *filename, const hr_analysis *a, char *buf, size_t cap)
try {
    if (!filename || !a) return HR_ERR_INVALID;
    char num[64];
    std::string o;
    csv_field(o, filename);
    o += ',';
    o += format_label(a->gain_method);
    o += ',';
    if (a->has_bitrate) { std::snprintf(num, sizeof num, "%u", a->bitrate_kbps); o += num; } else o += '-';
    // Rust's {:.1} / {:+.1} and C's %.1f / %+.1f both round the exact binary value to nearest, ties to even
    std::snprintf(num, sizeof num, ",%.1f,%.1f,%.1f,%+.1f,", a->input_i, a->input_tp, a->target_tp, a->headroom);
    o += num;
    o += method_label(a->gain_method);
    std::snprintf(num, sizeof num, ",%+.1f\n", a->effective_gain);
    o += num;
    return emit(o, buf, cap);
} HR_CATCH_ALL
