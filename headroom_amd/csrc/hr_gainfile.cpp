// hr_gainfile.cpp -- lossless gain for a WAV / AIFF file, in place: the file-level mirror of apply_gain_ffmpeg
// (src/processor.rs:53-87): `ffmpeg -y -i in -af volume={g}dB -c:a pcm_s24le -write_bext 1 tmp` (WAV) or
// `-c:a pcm_s24be -write_id3v2 1 tmp` (AIFF), then rename(tmp, in).  The sample arithmetic is hr_apply_gain's (GPU);
// this file only moves bytes: it rewrites the format chunk for 24-bit PCM, replaces the sound chunk and copies every
// other chunk through in its original place (the reference passes -write_bext / -write_id3v2 for exactly that
// purpose: keep bext, ID3 and the like).  Built on the public C ABI only.
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/headroom_hip.h"

// No exception crosses the C ABI: every int-returning entry point is a function-try-block (std::bad_alloc from a host
// container -> HR_ERR_OOM, anything else -> HR_ERR_INVALID).
#ifndef HR_CATCH_ALL
#define HR_CATCH_ALL catch (const std::bad_alloc &) { return HR_ERR_OOM; } catch (...) { return HR_ERR_INVALID; }
#endif

namespace {

void put_le16(std::vector<uint8_t> &o, uint32_t v) { o.push_back(v & 0xff); o.push_back((v >> 8) & 0xff); }
void put_le32(std::vector<uint8_t> &o, uint32_t v) { for (int i = 0; i < 4; i++) o.push_back((v >> (8 * i)) & 0xff); }
void put_be16(std::vector<uint8_t> &o, uint32_t v) { o.push_back((v >> 8) & 0xff); o.push_back(v & 0xff); }
void put_be32(std::vector<uint8_t> &o, uint32_t v) { for (int i = 3; i >= 0; i--) o.push_back((v >> (8 * i)) & 0xff); }
uint32_t get_le32(const uint8_t *p) { return p[0] | p[1] << 8 | p[2] << 16 | static_cast<uint32_t>(p[3]) << 24; }
uint32_t get_be32(const uint8_t *p) { return static_cast<uint32_t>(p[0]) << 24 | p[1] << 16 | p[2] << 8 | p[3]; }

// file_path.with_extension(format!("tmp.{}", extension)), src/processor.rs:58
std::string temp_path(const std::string &path)
{
    const size_t slash = path.find_last_of('/');
    const size_t dot = path.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash) || dot + 1 == path.size())
        return path + ".tmp.wav";  // (unwrap_or("wav"))
    return path.substr(0, dot) + ".tmp." + path.substr(dot + 1);
}

bool read_file(const char *path, std::vector<uint8_t> &out)
{
    std::FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0) { std::fclose(f); return false; }
    out.resize(static_cast<size_t>(n));
    const size_t got = n ? std::fread(out.data(), 1, out.size(), f) : 0;
    std::fclose(f);
    return got == out.size();
}

}  // namespace

extern "C" int hr_pcm_gain_file(hr_ctx *ctx, const char *path, double gain_db, uint64_t *clipped)
try {
    if (!ctx || !path) return HR_ERR_INVALID;
    std::vector<uint8_t> file;
    if (!read_file(path, file)) return HR_ERR_INVALID;
    hr_pcm_info info;
    int rc = hr_pcm_probe(file.data(), file.size(), &info);
    if (rc) return rc;
    const bool aiff = !std::memcmp(file.data(), "FORM", 4);
    void *pcm = nullptr;
    hr_track track;
    float weights[HR_MAX_CH];
    rc = hr_pcm_load_file(path, &pcm, &track, weights);
    if (rc) return rc;
    const uint64_t n_bytes = track.frames * track.channels * 3;
    // classic RIFF / IFF sizes are 32-bit (ffmpeg would switch to RF64 for WAV; not done here)
    if (n_bytes + file.size() > 0xfffffff0ull) { hr_free_pinned(pcm); return HR_ERR_TOO_LARGE; }
    std::vector<uint8_t> samples(n_bytes ? n_bytes : 1);
    rc = hr_apply_gain(ctx, &track, gain_db, aiff ? 1 : 0, samples.data(), clipped);
    hr_free_pinned(pcm);
    if (rc) return rc;
    samples.resize(n_bytes);

    std::vector<uint8_t> out;
    out.reserve(file.size() + n_bytes);
    const uint32_t block = track.channels * 3u;
    if (!aiff) {
        out.insert(out.end(), {'R', 'I', 'F', 'F', 0, 0, 0, 0, 'W', 'A', 'V', 'E'});
        size_t pos = 12;
        bool wrote_data = false;
        while (pos + 8 <= file.size()) {
            const uint8_t *ck = file.data() + pos;
            uint64_t size = get_le32(ck + 4);
            const size_t body = pos + 8;
            if (!std::memcmp(ck, "fmt ", 4)) {
                // WAVE_FORMAT_EXTENSIBLE when the source was (keeps its channel mask) or has more than two channels
                const bool ext = info.channel_mask != 0 || track.channels > 2;
                out.insert(out.end(), {'f', 'm', 't', ' '});
                put_le32(out, ext ? 40 : 16);
                put_le16(out, ext ? 0xFFFE : 1);
                put_le16(out, track.channels);
                put_le32(out, track.sample_rate);
                put_le32(out, track.sample_rate * block);
                put_le16(out, block);
                put_le16(out, 24);
                if (ext) {
                    put_le16(out, 22);
                    put_le16(out, 24);
                    put_le32(out, info.channel_mask);
                    static const uint8_t guid[16] = {1, 0, 0, 0, 0, 0, 0x10, 0, 0x80, 0, 0, 0xaa, 0, 0x38, 0x9b, 0x71};  // KSDATAFORMAT_SUBTYPE_PCM
                    out.insert(out.end(), guid, guid + 16);
                }
            } else if (!std::memcmp(ck, "data", 4)) {
                if (size == 0 || size == 0xFFFFFFFFu || body + size > file.size()) size = file.size() - body;
                if (!wrote_data) {
                    out.insert(out.end(), {'d', 'a', 't', 'a'});
                    put_le32(out, static_cast<uint32_t>(n_bytes));
                    out.insert(out.end(), samples.begin(), samples.end());
                    if (n_bytes & 1) out.push_back(0);
                    wrote_data = true;
                }
            } else if (!std::memcmp(ck, "fact", 4)) {
                // only meaningful for non-PCM data: dropped
            } else {
                const uint64_t take = std::min<uint64_t>(8 + size + (size & 1), file.size() - pos);
                out.insert(out.end(), ck, ck + take);
                if (take & 1) out.push_back(0);
            }
            pos = body + size + (size & 1);
        }
        if (!wrote_data) return HR_ERR_FORMAT;
        const uint32_t riff = static_cast<uint32_t>(out.size() - 8);
        for (int i = 0; i < 4; i++) out[4 + i] = (riff >> (8 * i)) & 0xff;
    } else {
        out.insert(out.end(), {'F', 'O', 'R', 'M', 0, 0, 0, 0, 'A', 'I', 'F', 'F'});  // AIFF-C input becomes plain AIFF (pcm_s24be)
        size_t pos = 12;
        bool wrote_ssnd = false;
        while (pos + 8 <= file.size()) {
            const uint8_t *ck = file.data() + pos;
            const uint64_t size = get_be32(ck + 4);
            const size_t body = pos + 8;
            if (!std::memcmp(ck, "COMM", 4)) {
                // (the probe has checked the COMM chunk it used; a file may carry a second, truncated one)
                if (body + 18 > file.size()) return HR_ERR_FORMAT;
                out.insert(out.end(), {'C', 'O', 'M', 'M'});
                put_be32(out, 18);
                put_be16(out, track.channels);
                put_be32(out, static_cast<uint32_t>(track.frames));
                put_be16(out, 24);
                out.insert(out.end(), file.data() + body + 8, file.data() + body + 18);  // the 80-bit sample rate, as it was
            } else if (!std::memcmp(ck, "SSND", 4)) {
                if (!wrote_ssnd) {
                    out.insert(out.end(), {'S', 'S', 'N', 'D'});
                    put_be32(out, static_cast<uint32_t>(n_bytes + 8));
                    put_be32(out, 0);
                    put_be32(out, 0);
                    out.insert(out.end(), samples.begin(), samples.end());
                    if (n_bytes & 1) out.push_back(0);
                    wrote_ssnd = true;
                }
            } else if (!std::memcmp(ck, "FVER", 4)) {
                // AIFF-C version stamp: not part of a plain AIFF file
            } else {
                const uint64_t take = std::min<uint64_t>(8 + size + (size & 1), file.size() - pos);
                out.insert(out.end(), ck, ck + take);
                if (take & 1) out.push_back(0);
            }
            pos = body + size + (size & 1);
        }
        if (!wrote_ssnd) return HR_ERR_FORMAT;
        const uint32_t form = static_cast<uint32_t>(out.size() - 8);
        for (int i = 0; i < 4; i++) out[4 + i] = (form >> (8 * (3 - i))) & 0xff;
    }

    // temp file next to the original, then rename over it: the original is never half-written (src/processor.rs:58,80-86)
    const std::string tmp = temp_path(path);
    std::FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return HR_ERR_INVALID;
    const size_t wr = std::fwrite(out.data(), 1, out.size(), f);
    const bool ok = std::fclose(f) == 0 && wr == out.size();
    if (!ok || std::rename(tmp.c_str(), path) != 0) {
        std::remove(tmp.c_str());
        return HR_ERR_INVALID;
    }
    return HR_OK;
} HR_CATCH_ALL
