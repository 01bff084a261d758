// hr_ingest.cpp -- host-side PCM container ingest: RIFF/WAVE and AIFF/AIFF-C headers -> hr_track.
//
// SURVEY.md 8(f) rank 1.  The reference hands every file to `ffmpeg -i <path> -map 0:a:0` (src/analyzer.rs:253-259)
// and accepts wav / aiff / aif among its lossless extensions (src/scanner.rs:6); for those two containers the
// first audio stream is plain PCM and needs no codec, so the meter can take the file as it lies on disk.
// Compressed formats (flac, mp3, aac) stay with the caller's decoder.
//
// What is reproduced from FFmpeg's demuxers (libavformat/wavdec.c, aiffdec.c -- published behaviour, not in
// the reference tree): chunk walk with even padding, WAVE_FORMAT_PCM / IEEE_FLOAT / EXTENSIBLE (sub-format
// from the GUID's first two bytes, channel mask kept), a `data` size of 0 / 0xFFFFFFFF or one that overruns
// the file means "to the end of the file"; AIFF COMM with the 80-bit extended sample rate, SSND offset,
// AIFF-C compression types NONE / twos (big endian), sowt (little endian), fl32 / FL32 / fl64 / FL64.
// 8-bit PCM is not a format the meter takes (HR_ERR_FORMAT).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/headroom_hip.h"
#include "hr_tables.hpp"

// No exception crosses the C ABI: every int-returning entry point is a function-try-block (std::bad_alloc from a host
// container -> HR_ERR_OOM, anything else -> HR_ERR_INVALID).
#ifndef HR_CATCH_ALL
#define HR_CATCH_ALL catch (const std::bad_alloc &) { return HR_ERR_OOM; } catch (...) { return HR_ERR_INVALID; }
#endif

namespace {

uint16_t le16(const uint8_t *p) { return static_cast<uint16_t>(p[0] | p[1] << 8); }
uint32_t le32(const uint8_t *p) { return static_cast<uint32_t>(p[0]) | static_cast<uint32_t>(p[1]) << 8 | static_cast<uint32_t>(p[2]) << 16 | static_cast<uint32_t>(p[3]) << 24; }
uint16_t be16(const uint8_t *p) { return static_cast<uint16_t>(p[0] << 8 | p[1]); }
uint32_t be32(const uint8_t *p) { return static_cast<uint32_t>(p[0]) << 24 | static_cast<uint32_t>(p[1]) << 16 | static_cast<uint32_t>(p[2]) << 8 | static_cast<uint32_t>(p[3]); }

// IEEE 754 80-bit extended, big endian (AIFF COMM sampleRate)
double be_extended(const uint8_t *p)
{
    const int sign = p[0] >> 7;
    const int exp = ((p[0] & 0x7f) << 8) | p[1];
    uint64_t mant = 0;
    for (int i = 0; i < 8; i++) mant = mant << 8 | p[2 + i];
    if (exp == 0 && mant == 0) return 0.0;
    const double v = std::ldexp(static_cast<double>(mant), exp - 16383 - 63);
    return sign ? -v : v;
}

int pcm_fmt(bool is_float, int bits)
{
    if (is_float) return bits == 32 ? HR_FMT_F32 : bits == 64 ? HR_FMT_F64 : -1;
    return bits == 16 ? HR_FMT_S16 : bits == 24 ? HR_FMT_S24 : bits == 32 ? HR_FMT_S32 : -1;
}

int bytes_of(int fmt) { return fmt == HR_FMT_S16 ? 2 : fmt == HR_FMT_S24 ? 3 : fmt == HR_FMT_F64 ? 8 : 4; }

// b[0..n) is a prefix of a file of `total` bytes: chunk headers must lie inside the prefix, sample data need not
int probe_wav(const uint8_t *b, size_t n, uint64_t total, hr_pcm_info *o)
{
    size_t pos = 12;
    bool have_fmt = false;
    while (pos + 8 <= n) {
        const uint8_t *ck = b + pos;
        const uint32_t size = le32(ck + 4);
        const size_t body = pos + 8;
        if (!std::memcmp(ck, "fmt ", 4)) {
            if (size < 16 || body + 16 > n) return HR_ERR_FORMAT;
            int tag = le16(b + body);
            o->channels = le16(b + body + 2);
            o->sample_rate = le32(b + body + 4);
            if (o->sample_rate == 0 || o->sample_rate > HR_MAX_SAMPLE_RATE) return HR_ERR_RATE;
            const int bits = le16(b + body + 14);
            o->channel_mask = 0;
            if (tag == 0xFFFE) {
                if (size < 40 || body + 40 > n) return HR_ERR_FORMAT;
                o->channel_mask = le32(b + body + 20);
                tag = le16(b + body + 24);  // first two bytes of the sub-format GUID
            }
            if (tag != 1 && tag != 3) return HR_ERR_FORMAT;  // compressed WAV: not PCM
            const int f = pcm_fmt(tag == 3, bits);
            if (f < 0 || o->channels < 1 || o->channels > HR_MAX_CH) return HR_ERR_FORMAT;
            o->fmt = static_cast<uint16_t>(f);
            o->big_endian = 0;
            have_fmt = true;
        } else if (!std::memcmp(ck, "data", 4)) {
            if (!have_fmt) return HR_ERR_FORMAT;
            uint64_t len = size;
            if (size == 0 || size == 0xFFFFFFFFu || body + len > total) len = total - body;  // streamed / truncated / >4 GiB files
            o->data_offset = body;
            o->frames = len / (static_cast<uint64_t>(bytes_of(o->fmt)) * o->channels);
            return HR_OK;
        }
        pos = body + size + (size & 1);
    }
    return HR_ERR_FORMAT;
}

int probe_aiff(const uint8_t *b, size_t n, uint64_t total, hr_pcm_info *o)
{
    const bool aifc = !std::memcmp(b + 8, "AIFC", 4);
    size_t pos = 12;
    bool have_comm = false, have_ssnd = false;
    uint64_t comm_frames = 0, ssnd_off = 0, ssnd_len = 0;
    while (pos + 8 <= n) {
        const uint8_t *ck = b + pos;
        const uint32_t size = be32(ck + 4);
        const size_t body = pos + 8;
        if (!std::memcmp(ck, "COMM", 4)) {
            if (size < 18 || body + 18 > n) return HR_ERR_FORMAT;
            o->channels = be16(b + body);
            comm_frames = be32(b + body + 2);
            const int bits = be16(b + body + 6);
            const double rate = be_extended(b + body + 8);
            if (!(rate >= 1.0 && rate <= static_cast<double>(HR_MAX_SAMPLE_RATE))) return HR_ERR_RATE;
            o->sample_rate = static_cast<uint32_t>(std::llround(rate));
            bool is_float = false;
            o->big_endian = 1;
            if (aifc) {
                if (size < 22 || body + 22 > n) return HR_ERR_FORMAT;
                const uint8_t *ct = b + body + 18;
                if (!std::memcmp(ct, "NONE", 4) || !std::memcmp(ct, "twos", 4)) {
                } else if (!std::memcmp(ct, "sowt", 4)) {
                    o->big_endian = 0;
                } else if (!std::memcmp(ct, "fl32", 4) || !std::memcmp(ct, "FL32", 4) || !std::memcmp(ct, "fl64", 4) || !std::memcmp(ct, "FL64", 4)) {
                    is_float = true;
                } else {
                    return HR_ERR_FORMAT;  // compressed AIFF-C
                }
            }
            const int f = pcm_fmt(is_float, is_float ? bits : (bits + 7) / 8 * 8);
            if (f < 0 || o->channels < 1 || o->channels > HR_MAX_CH) return HR_ERR_FORMAT;
            o->fmt = static_cast<uint16_t>(f);
            o->channel_mask = 0;
            have_comm = true;
        } else if (!std::memcmp(ck, "SSND", 4)) {
            if (size < 8 || body + 8 > n) return HR_ERR_FORMAT;
            const uint32_t offset = be32(b + body);
            ssnd_off = body + 8 + offset;
            ssnd_len = size >= 8 + static_cast<uint64_t>(offset) ? size - 8 - offset : 0;
            if (ssnd_off > total) return HR_ERR_FORMAT;
            if (ssnd_off + ssnd_len > total) ssnd_len = total - ssnd_off;
            have_ssnd = true;
        }
        pos = body + size + (size & 1);
    }
    if (!have_comm || !have_ssnd) return HR_ERR_FORMAT;
    const uint64_t fb = static_cast<uint64_t>(bytes_of(o->fmt)) * o->channels;
    o->data_offset = ssnd_off;
    o->frames = std::min<uint64_t>(comm_frames, ssnd_len / fb);
    return HR_OK;
}

}  // namespace

extern "C" {

static int probe_any(const uint8_t *b, size_t n, uint64_t total, hr_pcm_info *info)
{
    std::memset(info, 0, sizeof *info);
    if (n < 12) return HR_ERR_FORMAT;
    if (!std::memcmp(b, "RIFF", 4) && !std::memcmp(b + 8, "WAVE", 4)) return probe_wav(b, n, total, info);
    if (!std::memcmp(b, "FORM", 4) && (!std::memcmp(b + 8, "AIFF", 4) || !std::memcmp(b + 8, "AIFC", 4))) return probe_aiff(b, n, total, info);
    return HR_ERR_FORMAT;
}

int hr_pcm_probe(const void *file_bytes, size_t n, hr_pcm_info *info)
try {
    if (!file_bytes || !info) return HR_ERR_INVALID;
    return probe_any(static_cast<const uint8_t *>(file_bytes), n, n, info);
} HR_CATCH_ALL

int hr_pcm_load_file(const char *path, void **pcm_out, hr_track *track, float *weights)
try {
    if (!path || !pcm_out || !track || !weights) return HR_ERR_INVALID;
    *pcm_out = nullptr;
    std::FILE *f = std::fopen(path, "rb");
    if (!f) return HR_ERR_INVALID;
    std::fseek(f, 0, SEEK_END);
    const long long file_size = std::ftell(f);
    if (file_size < 0) { std::fclose(f); return HR_ERR_INVALID; }
    // the header chunks sit in front of the samples in any file a recorder writes: probe a 1 MiB prefix, and the
    // whole file only if the sound chunk was not found in it (AIFF allows SSND before COMM, WAV a long LIST first)
    std::vector<uint8_t> head(std::min<uint64_t>(1 << 20, static_cast<uint64_t>(file_size)));
    std::fseek(f, 0, SEEK_SET);
    size_t got = std::fread(head.data(), 1, head.size(), f);
    hr_pcm_info info;
    int rc = probe_any(head.data(), got, static_cast<uint64_t>(file_size), &info);
    if (rc != HR_OK && static_cast<uint64_t>(file_size) > got) {
        head.resize(static_cast<size_t>(file_size));
        std::fseek(f, 0, SEEK_SET);
        got = std::fread(head.data(), 1, head.size(), f);
        rc = probe_any(head.data(), got, static_cast<uint64_t>(file_size), &info);
    }
    if (rc != HR_OK) { std::fclose(f); return rc; }
    const uint64_t fb = static_cast<uint64_t>(bytes_of(info.fmt)) * info.channels;
    const uint64_t bytes = info.frames * fb;
    void *buf = hr_alloc_pinned(bytes ? bytes : 1);
    if (!buf) { std::fclose(f); return HR_ERR_OOM; }
    std::fseek(f, static_cast<long>(info.data_offset), SEEK_SET);
    const size_t rd = bytes ? std::fread(buf, 1, bytes, f) : 0;
    std::fclose(f);
    if (rd != bytes) { hr_free_pinned(buf); return HR_ERR_INVALID; }
    if (info.big_endian) {  // the meter's formats are little endian
        const int w = bytes_of(info.fmt);
        uint8_t *p = static_cast<uint8_t *>(buf);
        for (uint64_t i = 0; i + w <= bytes; i += w)
            for (int a = 0, z = w - 1; a < z; a++, z--) std::swap(p[i + a], p[i + z]);
    }
    if (info.channel_mask) hr::weights_from_mask(info.channel_mask, info.channels, weights);
    else hr::default_weights(info.channels, weights);
    std::memset(track, 0, sizeof *track);
    track->pcm = buf;
    track->frames = info.frames;
    track->sample_rate = info.sample_rate;
    track->channels = info.channels;
    track->fmt = info.fmt;
    track->ch_weight = weights;
    *pcm_out = buf;
    return HR_OK;
} HR_CATCH_ALL

}  // extern "C"
