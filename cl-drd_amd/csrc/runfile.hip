// Run-file writer of the retrieve path (reference retriever/retrieve_top_passages.py:98-105):
//     f.write(f"{qid}\t{docid}\t{i+1}\t{s}\n")       for every query in encode order, rank i+1 = 1..k, s = a Python float
// s comes from `batch_nn_scores.tolist()` (retriever/retrieval_utils.py:146): the fp32 score widened to a double, printed by Python's
// float repr = the SHORTEST decimal string that round-trips the double, in fixed notation for 1e-4 <= |s| < 1e16 (always with a decimal
// point) and d.ddde+XX otherwise.  6980 queries x 1000 hits are 6.98 M lines: the Python loop takes seconds next to a 20-ms search, so the
// lines are formatted here, host-side native code, on all host cores: std::to_chars gives the shortest round-trip digits, the notation
// rule is restated from CPython's float_repr (PyOS_double_to_string 'r': exponent form iff decpt > 16 or decpt < -3).
// No GPU work in this file.
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

int cldrd_set_error(const char* msg);

namespace {

// Python repr(float(x)) into out (>= 32 bytes); returns the length
inline int py_float_repr(double x, char* out) {
    if (std::isnan(x)) { memcpy(out, "nan", 3); return 3; }
    if (std::isinf(x)) { if (x < 0) { memcpy(out, "-inf", 4); return 4; } memcpy(out, "inf", 3); return 3; }
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof(sci), x, std::chars_format::scientific);     // [-]d[.ddd]e[+-]XX, shortest round trip
    const char* p = sci;
    int n = 0;
    if (*p == '-') { out[n++] = '-'; ++p; }
    char digits[24];
    int nd = 0;
    const char* e = p;
    while (e < r.ptr && *e != 'e') { if (*e != '.') digits[nd++] = *e; ++e; }
    int exp10 = 0;
    {
        const char* q = e + 1;
        const bool neg = *q == '-';
        if (*q == '-' || *q == '+') ++q;
        while (q < r.ptr) exp10 = exp10 * 10 + (*q++ - '0');
        if (neg) exp10 = -exp10;
    }
    while (nd > 1 && digits[nd - 1] == '0') --nd;                  // to_chars never pads, but 0 prints as "0e+00"
    const int decpt = exp10 + 1;                                   // value = 0.d1d2... x 10^decpt
    if (decpt > 16 || decpt < -3) {                                // exponent form: d[.ddd]e[+-]XX (at least two exponent digits)
        out[n++] = digits[0];
        if (nd > 1) { out[n++] = '.'; memcpy(out + n, digits + 1, nd - 1); n += nd - 1; }
        out[n++] = 'e';
        int ex = decpt - 1;
        out[n++] = ex < 0 ? '-' : '+';
        if (ex < 0) ex = -ex;
        char eb[8];
        int ne = 0;
        do { eb[ne++] = (char)('0' + ex % 10); ex /= 10; } while (ex);
        if (ne < 2) eb[ne++] = '0';
        while (ne) out[n++] = eb[--ne];
        return n;
    }
    if (decpt <= 0) {                                              // 0.000ddd
        out[n++] = '0'; out[n++] = '.';
        for (int i = 0; i < -decpt; ++i) out[n++] = '0';
        memcpy(out + n, digits, nd); n += nd;
        return n;
    }
    if (nd <= decpt) {                                             // integer value: ddd000.0
        memcpy(out + n, digits, nd); n += nd;
        for (int i = nd; i < decpt; ++i) out[n++] = '0';
        out[n++] = '.'; out[n++] = '0';
        return n;
    }
    memcpy(out + n, digits, decpt); n += decpt;
    out[n++] = '.';
    memcpy(out + n, digits + decpt, nd - decpt); n += nd - decpt;
    return n;
}

inline int put_i64(long long v, char* out) {
    char b[24];
    int nb = 0, n = 0;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { b[nb++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) out[n++] = '-';
    while (nb) out[n++] = b[--nb];
    return n;
}

}  // namespace

// repr of one value (tests pin the formatter against Python's repr on a few million floats); returns the length written to out[32]
extern "C" int cldrd_py_float_repr(double x, char* out) { return py_float_repr(x, out); }

// Formats nq x k lines into `path` (truncating it).  qids int64[nq]; docids int64[nq, k]; scores fp32[nq, k]; all on the HOST.
// Returns the number of lines written, or -1 (cldrd_last_error()).  nthreads <= 0: one per hardware thread, at most 64.
extern "C" long long cldrd_write_run_file(const char* path, const long long* qids, const long long* docids, const float* scores,
                                          long long nq, int k, int nthreads) {
    if (!path || nq < 0 || k <= 0 || (nq > 0 && (!qids || !docids || !scores))) { cldrd_set_error("write_run_file: bad arguments"); return -1; }
    int nt = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    if ((long long)nt > nq) nt = nq > 0 ? (int)nq : 1;
    std::vector<std::string> parts((size_t)nt);
    auto work = [&](int t) {
        const long long lo = nq * t / nt, hi = nq * (t + 1) / nt;
        std::string& s = parts[(size_t)t];
        s.resize((size_t)(hi - lo) * k * 80);                     // qid (<= 20) + docid (<= 20) + rank (<= 10) + score (<= 25) + 4 separators
        char* o = &s[0];
        for (long long q = lo; q < hi; ++q) {
            char qb[24];
            const int nqb = put_i64(qids[q], qb);
            for (int i = 0; i < k; ++i) {
                memcpy(o, qb, nqb); o += nqb; *o++ = '\t';
                o += put_i64(docids[q * k + i], o); *o++ = '\t';
                o += put_i64(i + 1, o); *o++ = '\t';
                o += py_float_repr((double)scores[q * k + i], o); *o++ = '\n';
            }
        }
        s.resize((size_t)(o - &s[0]));
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    FILE* f = fopen(path, "wb");
    if (!f) { cldrd_set_error("write_run_file: cannot open the output file"); return -1; }
    for (const auto& s : parts)
        if (!s.empty() && fwrite(s.data(), 1, s.size(), f) != s.size()) { fclose(f); cldrd_set_error("write_run_file: short write"); return -1; }
    if (fclose(f) != 0) { cldrd_set_error("write_run_file: close failed"); return -1; }
    return nq * k;
}
