// Input pipeline of the training step: native parser of the reference's on-disk alignment pairs
// (`<n>_sites.csv`: 64 lines of 64 comma-separated 0/1, written by src/sim_conversion/convert_ms.py:39;
// `<n>_pos.csv`: 64 lines of one float, convert_ms.py:28-29 / convert_relative.py:70-73) into a bit-packed
// slab (512 bytes of genotype bits + 64 fp32 positions per alignment, 768 B instead of 16.6 KB), and the
// device kernel that expands packed alignments into the fp32 pool the step reads (with the
// `--normalize` map (x - 0.5) / 0.5 of src/train_wgan_v2.py:453,519-521 fused).
// Replaces the 2 B np.loadtxt calls per batch of DataGeneratorDisk.get_batch (src/train_wgan_v2.py:254-255).
#include "pgan_common.h"

#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int ROWS = 64, COLS = 64, BITS_BYTES = ROWS * COLS / 8;

// whole file into a buffer (the files are ~8 KB); false if it cannot be read
bool slurp(const char* path, std::vector<char>& buf) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    buf.clear();
    char tmp[16384];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(f);
    buf.push_back('\0');
    return true;
}

// 0: ok; 1: unreadable; 2: not ROWS x COLS values; 3: a value other than 0 or 1
int parse_sites(const char* path, uint8_t* bits, std::vector<char>& buf) {
    if (!slurp(path, buf)) return 1;
    std::memset(bits, 0, BITS_BYTES);
    int row = 0, col = 0;
    const char* p = buf.data();
    while (*p) {
        while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
        if (*p == '\n') {
            if (col) { if (col != COLS) return 2; ++row; col = 0; }
            ++p;
            continue;
        }
        if (!*p) break;
        char* end;
        const double v = std::strtod(p, &end);     // np.loadtxt accepts "1", "1.0", "1e0": so does this
        if (end == p) return 2;
        if (v != 0.0 && v != 1.0) return 3;
        if (row >= ROWS || col >= COLS) return 2;
        if (v == 1.0) bits[row * (COLS / 8) + col / 8] |= (uint8_t)(1u << (col % 8));
        ++col;
        p = end;
        while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
        if (*p == ',') ++p;
    }
    if (col) { if (col != COLS) return 2; ++row; }
    return row == ROWS ? 0 : 2;
}

// one float per line; parsed as a double and rounded once to fp32, as np.loadtxt(dtype=float32) does
int parse_pos(const char* path, float* pos, std::vector<char>& buf) {
    if (!slurp(path, buf)) return 1;
    int n = 0;
    const char* p = buf.data();
    while (*p) {
        while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == ',') ++p;
        if (!*p) break;
        char* end;
        const double v = std::strtod(p, &end);
        if (end == p || n >= COLS) return 2;
        pos[n++] = (float)v;
        p = end;
    }
    return n == COLS ? 0 : 2;
}

// out[n][64][64] = scale * bit + shift; one thread expands one byte (8 sites) into two float4 stores
__global__ __launch_bounds__(256) void unpack_sites_kernel(const uint8_t* __restrict__ bits, float* __restrict__ out,
                                                           int64_t nbytes, float scale, float shift) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nbytes) return;
    const uint32_t b = bits[i];
    f32x4 lo, hi;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        lo[e] = ((b >> e) & 1u) ? scale + shift : shift;
        hi[e] = ((b >> (e + 4)) & 1u) ? scale + shift : shift;
    }
    f32x4* o = reinterpret_cast<f32x4*>(out + i * 8);
    o[0] = lo;
    o[1] = hi;
}

// bits[n][512] from out-style fp32 {0,1} alignments (x >= 0.5 -> 1): the inverse, for packing generated data
__global__ __launch_bounds__(256) void pack_sites_kernel(const float* __restrict__ x, uint8_t* __restrict__ bits,
                                                         int64_t nbytes) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nbytes) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(x + i * 8);
    const f32x4 lo = p[0], hi = p[1];
    uint32_t b = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        b |= (lo[e] >= 0.5f ? 1u : 0u) << e;
        b |= (hi[e] >= 0.5f ? 1u : 0u) << (e + 4);
    }
    bits[i] = (uint8_t)b;
}

}  // namespace

extern "C" {

// Host side (no GPU involved).  n alignment pairs -> bits[n][512] (row-major, site j of individual i is bit j % 8 of
// byte i * 8 + j / 8) and pos[n][64].  status[k]: 0 ok, 1 file unreadable, 2 wrong number of values, 3 a site value
// other than 0 / 1 (label-smoothed files are not packable).  Returns the number of failed pairs, -1 on bad arguments.
int pgan_parse_alignments(const char* const* sites_paths, const char* const* pos_paths, int n, unsigned char* bits,
                          float* pos, int* status, int threads) {
    if (n < 0 || !sites_paths || !pos_paths || !bits || !pos || !status) return PGAN_EINVAL;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    std::atomic<int> next(0), failed(0);
    auto work = [&]() {
        std::vector<char> buf;
        for (;;) {
            const int k = next.fetch_add(1);
            if (k >= n) break;
            int rc = parse_sites(sites_paths[k], bits + (size_t)k * BITS_BYTES, buf);
            if (rc == 0) rc = parse_pos(pos_paths[k], pos + (size_t)k * COLS, buf);
            status[k] = rc;
            if (rc) failed.fetch_add(1);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads && t < n; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    return failed.load();
}

// Host side.  Text output of Hudson's `ms` (what SimulatorGenerator.get_batch reads from the subprocess,
// src/train_wgan_v2.py:342-344) -> sites[r][rows][cols] (0 / 1 as fp32) and pos[r][cols], following read_ms_data
// (src/train_wgan_v2.py:311-330) line for line: the text is split at '\n'; a replicate starts at every line that
// contains "//"; its third line is "positions: ..." split at single spaces with the first and the LAST token dropped
// (ms ends that line with a space) and empty tokens skipped; the lines after it up to, not including, the last line
// before the next "//" are one individual each, one digit per site; the LAST replicate of the text is dropped (the
// caller asks ms for one more than it needs).  Returns the number of replicates written (<= max_reps), or
// -1 bad arguments, -3 a replicate with other than rows x cols genotypes or cols positions, -4 a non-digit genotype.
int pgan_parse_ms(const char* text, long long len, int rows, int cols, int max_reps, float* sites, float* pos) {
    if (!text || len < 0 || rows <= 0 || cols <= 0 || max_reps < 0 || !sites || !pos) return PGAN_EINVAL;
    std::vector<std::pair<const char*, const char*>> lines;      // [begin, end) of every '\n'-separated piece
    const char* b = text;
    const char* const fin = text + len;
    for (const char* p = text;; ++p) {
        if (p == fin || *p == '\n') {
            lines.emplace_back(b, p);
            if (p == fin) break;
            b = p + 1;
        }
    }
    std::vector<size_t> starts;
    for (size_t i = 0; i < lines.size(); ++i) {
        for (const char* p = lines[i].first; p + 1 < lines[i].second; ++p)
            if (p[0] == '/' && p[1] == '/') { starts.push_back(i); break; }
    }
    if (starts.empty()) return 0;
    starts.push_back(lines.size());
    int nrep = 0;
    for (size_t k = 0; k + 2 < starts.size() && nrep < max_reps; ++k, ++nrep) {      // all chunks but the last
        const size_t l0 = starts[k], l1 = starts[k + 1];
        if (l1 < l0 + 4) return -3;
        // positions: tokens between single spaces, first and last dropped, empty ones skipped
        std::vector<std::pair<const char*, const char*>> tok;
        const char* tb = lines[l0 + 2].first;
        const char* const te = lines[l0 + 2].second;
        for (const char* p = tb;; ++p) {
            if (p == te || *p == ' ') {
                tok.emplace_back(tb, p);
                if (p == te) break;
                tb = p + 1;
            }
        }
        int np = 0;
        for (size_t t = 1; t + 1 < tok.size(); ++t) {
            if (tok[t].first == tok[t].second) continue;
            if (np >= cols) return -3;
            const std::string v(tok[t].first, tok[t].second);
            char* end;
            const double d = std::strtod(v.c_str(), &end);
            if (end == v.c_str()) return -3;
            pos[(size_t)nrep * cols + np++] = (float)d;
        }
        if (np != cols) return -3;
        if ((long long)(l1 - 1) - (long long)(l0 + 3) != rows) return -3;
        for (int r = 0; r < rows; ++r) {
            const auto& ln = lines[l0 + 3 + r];
            if (ln.second - ln.first != cols) return -3;
            for (int c = 0; c < cols; ++c) {
                const char ch = ln.first[c];
                if (ch < '0' || ch > '9') return -4;
                sites[((size_t)nrep * rows + r) * cols + c] = (float)(ch - '0');
            }
        }
    }
    return nrep;
}

// Device side: sites[n][1][64][64] (fp32) = scale * bit + shift from bits[n][512].
int pgan_unpack_sites(const unsigned char* bits, float* sites, long long n, float scale, float shift, hipStream_t st) {
    if (n <= 0 || !bits || !sites) return PGAN_EINVAL;
    const int64_t nbytes = (int64_t)n * BITS_BYTES;
    unpack_sites_kernel<<<(unsigned)((nbytes + 255) / 256), 256, 0, st>>>(bits, sites, nbytes, scale, shift);
    return pgan_launch_status();
}

int pgan_pack_sites(const float* sites, unsigned char* bits, long long n, hipStream_t st) {
    if (n <= 0 || !bits || !sites) return PGAN_EINVAL;
    const int64_t nbytes = (int64_t)n * BITS_BYTES;
    pack_sites_kernel<<<(unsigned)((nbytes + 255) / 256), 256, 0, st>>>(sites, bits, nbytes);
    return pgan_launch_status();
}

}  // extern "C"
