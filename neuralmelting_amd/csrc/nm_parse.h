// nm_parse.h — reader of the consolidated .thrm / .traj text files (second half): the work of
// walkernr/neuralMelting's scripts/lammps_parse.py:48-49 (np.loadtxt) and :88-96 (readlines + split + per-line np.array),
// which slurps C5's 2 M lines per recorded cycle into Python lists.  The file is memory-mapped, cut into byte ranges at
// line boundaries, counted and then converted by a pool of threads.  Numbers are converted text -> double -> float32, the
// same two roundings numpy makes; '%.4E' tokens take an exact fast path (5-digit integer times/over an exact power of ten
// is one correctly rounded operation), anything else goes through strtod.
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace nm {

struct MappedFile {
    const char *p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open(const char *path, std::string &err)
    {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) { err = std::string("cannot open ") + path + ": " + std::strerror(errno); return false; }
        struct stat st;
        if (fstat(fd, &st) != 0) { err = std::string("cannot stat ") + path; return false; }
        n = (size_t)st.st_size;
        if (n == 0) return true;
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { err = std::string("cannot map ") + path + ": " + std::strerror(errno); return false; }
        madvise(m, n, MADV_SEQUENTIAL);
        p = (const char *)m;
        return true;
    }
    ~MappedFile()
    {
        if (p) munmap((void *)p, n);
        if (fd >= 0) ::close(fd);
    }
};

inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

// one token [s,e) -> double
inline double parse_number(const char *s, const char *e)
{
    static const double P10[23] = { 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                                    1e17, 1e18, 1e19, 1e20, 1e21, 1e22 };
    const char *q = s;
    bool neg = false;
    if (q < e && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
    // d.ddddE[+-]dd
    if (e - q == 10 && q[1] == '.' && (q[6] == 'E' || q[6] == 'e') && (q[7] == '+' || q[7] == '-')) {
        bool digits = true;
        const int idx[7] = { 0, 2, 3, 4, 5, 8, 9 };
        for (int k = 0; k < 7; ++k) digits = digits && q[idx[k]] >= '0' && q[idx[k]] <= '9';
        if (digits) {
            const long m = (q[0] - '0') * 10000L + (q[2] - '0') * 1000L + (q[3] - '0') * 100L + (q[4] - '0') * 10L + (q[5] - '0');
            const int ex = ((q[8] - '0') * 10 + (q[9] - '0')) * (q[7] == '-' ? -1 : 1) - 4;
            if (ex >= -22 && ex <= 22) {
                const double v = ex >= 0 ? (double)m * P10[ex] : (double)m / P10[-ex];
                return neg ? -v : v;
            }
        }
    }
    char buf[64];
    size_t len = (size_t)(e - s);
    if (len >= sizeof buf) len = sizeof buf - 1;
    std::memcpy(buf, s, len);
    buf[len] = 0;
    return std::strtod(buf, nullptr);
}

// tokens of one line; returns the count (at most cap are stored)
inline int split_line(const char *s, const char *e, const char **ts, const char **te, int cap)
{
    int n = 0;
    while (s < e) {
        while (s < e && is_blank(*s)) ++s;
        if (s >= e) break;
        const char *t = s;
        while (s < e && !is_blank(*s)) ++s;
        if (n < cap) { ts[n] = t; te[n] = s; }
        ++n;
    }
    return n;
}

// byte ranges [cut[i], cut[i+1]) that begin at line starts
inline std::vector<size_t> cut_at_lines(const char *p, size_t n, int parts)
{
    std::vector<size_t> cut(parts + 1, n);
    cut[0] = 0;
    for (int i = 1; i < parts; ++i) {
        size_t c = n / parts * i;
        if (c < cut[i - 1]) c = cut[i - 1];
        const void *nl = c < n ? std::memchr(p + c, '\n', n - c) : nullptr;
        cut[i] = nl ? (size_t)((const char *)nl - p) + 1 : n;
    }
    return cut;
}

template <class F>
inline void for_lines(const char *p, size_t b, size_t e, F &&f)
{
    while (b < e) {
        const void *nl = std::memchr(p + b, '\n', e - b);
        const size_t le = nl ? (size_t)((const char *)nl - p) : e;
        f(p + b, p + le);
        b = le + 1;
    }
}

inline int pick_threads(int nthreads, size_t bytes)
{
    int nt = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    const size_t by_size = bytes / (1u << 20) + 1; // a thread per MiB at most
    if ((size_t)nt > by_size) nt = (int)by_size;
    return nt;
}

// .thrm: every non-empty line that does not start with '#' holds 17 numbers.  rows may be null (count only).
inline int parse_thrm(const char *path, float *rows, long cap_rows, long *nrows, int nthreads, std::string &err)
{
    MappedFile mf;
    if (!mf.open(path, err)) return -1;
    const int nt = pick_threads(nthreads, mf.n);
    const std::vector<size_t> cut = cut_at_lines(mf.p, mf.n, nt);
    std::vector<long> count(nt, 0), bad(nt, 0);
    auto classify = [&](const char *s, const char *e, const char **ts, const char **te) -> int {
        while (s < e && is_blank(*s)) ++s;
        if (s >= e || *s == '#') return 0;
        const char *h = (const char *)std::memchr(s, '#', (size_t)(e - s)); // np.loadtxt strips trailing comments
        return split_line(s, h ? h : e, ts, te, 17);
    };
    {
        std::vector<std::thread> th;
        for (int i = 0; i < nt; ++i)
            th.emplace_back([&, i] {
                const char *ts[17], *te[17];
                for_lines(mf.p, cut[i], cut[i + 1], [&](const char *s, const char *e) {
                    const int k = classify(s, e, ts, te);
                    if (k == 17) ++count[i];
                    else if (k != 0) ++bad[i];
                });
            });
        for (auto &t : th) t.join();
    }
    long total = 0;
    for (int i = 0; i < nt; ++i) {
        if (bad[i]) { err = std::string(path) + ": a data line does not hold 17 columns"; return -1; }
        total += count[i];
    }
    if (nrows) *nrows = total;
    if (!rows) return 0;
    if (total > cap_rows) { err = "parse_thrm: output buffer too small"; return -1; }
    std::vector<long> first(nt, 0);
    for (int i = 1; i < nt; ++i) first[i] = first[i - 1] + count[i - 1];
    std::vector<std::thread> th;
    for (int i = 0; i < nt; ++i)
        th.emplace_back([&, i] {
            const char *ts[17], *te[17];
            float *o = rows + 17 * first[i];
            for_lines(mf.p, cut[i], cut[i + 1], [&](const char *s, const char *e) {
                if (classify(s, e, ts, te) != 17) return;
                for (int c = 0; c < 17; ++c) o[c] = (float)parse_number(ts[c], te[c]);
                o += 17;
            });
        });
    for (auto &t : th) t.join();
    return 0;
}

// .traj: lines of two tokens are frame heads (natoms, box), lines of three are coordinates, anything else is skipped
// (lammps_parse.py:90-93 filters by len(values)).  Outputs may be null (count only).
inline int parse_traj(const char *path, uint16_t *natoms, float *box, float *pos, long cap_frames, long cap_rows, long *nframes,
                      long *nposrows, int nthreads, std::string &err)
{
    MappedFile mf;
    if (!mf.open(path, err)) return -1;
    const int nt = pick_threads(nthreads, mf.n);
    const std::vector<size_t> cut = cut_at_lines(mf.p, mf.n, nt);
    std::vector<long> heads(nt, 0), coords(nt, 0);
    {
        std::vector<std::thread> th;
        for (int i = 0; i < nt; ++i)
            th.emplace_back([&, i] {
                const char *ts[4], *te[4];
                for_lines(mf.p, cut[i], cut[i + 1], [&](const char *s, const char *e) {
                    const int k = split_line(s, e, ts, te, 4);
                    if (k == 2) ++heads[i];
                    else if (k == 3) ++coords[i];
                });
            });
        for (auto &t : th) t.join();
    }
    long nh = 0, nc = 0;
    for (int i = 0; i < nt; ++i) { nh += heads[i]; nc += coords[i]; }
    if (nframes) *nframes = nh;
    if (nposrows) *nposrows = nc;
    if (!natoms && !box && !pos) return 0;
    if (nh > cap_frames || nc > cap_rows) { err = "parse_traj: output buffer too small"; return -1; }
    std::vector<long> h0(nt, 0), c0(nt, 0);
    for (int i = 1; i < nt; ++i) { h0[i] = h0[i - 1] + heads[i - 1]; c0[i] = c0[i - 1] + coords[i - 1]; }
    std::vector<std::thread> th;
    for (int i = 0; i < nt; ++i)
        th.emplace_back([&, i] {
            const char *ts[4], *te[4];
            long h = h0[i], c = c0[i];
            for_lines(mf.p, cut[i], cut[i + 1], [&](const char *s, const char *e) {
                const int k = split_line(s, e, ts, te, 4);
                if (k == 2) {
                    if (natoms) natoms[h] = (uint16_t)std::strtol(std::string(ts[0], te[0]).c_str(), nullptr, 10);
                    if (box) box[h] = (float)parse_number(ts[1], te[1]);
                    ++h;
                } else if (k == 3) {
                    if (pos)
                        for (int d = 0; d < 3; ++d) pos[3 * c + d] = (float)parse_number(ts[d], te[d]);
                    ++c;
                }
            });
        });
    for (auto &t : th) t.join();
    return 0;
}

} // namespace nm
