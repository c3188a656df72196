// fasta_io.cpp -- host-side FASTA reader and BED/TSV writers of libprf (no GPU code).
//
// The two data formats either side of the hot path:
//  * prf_fasta_*: what the reference gets from pyfastx.Fasta (reference perfect_repeat_finder.py:117,
//    :130, :136-143): entries in file order, name = header up to the first white space, sequence = the
//    record's lines joined, case preserved.  Plain or gzip-compressed text (zlib reads both).
//  * prf_write_bed / prf_write_tsv: the reference's output lines, "chrom\tstart\tend\tmotif\n"
//    (:148-149) and "start_0based\tend\tmotif" + rows (:166-170), motif = seq.upper()[start:start+k].
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <memory>
#include <string>
#include <vector>

#include "../../include/prf.h"

int prf_set_error(int code, const char *fmt, ...);  // api.cpp

// sequence bytes: malloc'ed, never zero-filled, always 8 bytes of slack behind size() for word-wide stores
struct prf_seq {
    char *p = nullptr;
    size_t n = 0, cap = 0;
    prf_seq() = default;
    prf_seq(prf_seq &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    prf_seq &operator=(prf_seq &&o) noexcept {
        if (this != &o) { free(p); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; }
        return *this;
    }
    prf_seq(const prf_seq &) = delete;
    prf_seq &operator=(const prf_seq &) = delete;
    ~prf_seq() { free(p); }
    void reserve(size_t want) {
        if (want <= cap) return;
        char *q = (char *)realloc(p, want + 8);
        if (!q) throw std::bad_alloc();
        p = q;
        cap = want;
    }
    void room(size_t more) {
        if (n + more > cap) reserve(std::max(n + more, cap + cap / 2 + 4096));
    }
    const char *data() const { return p; }
    size_t size() const { return n; }
};

struct prf_fasta {
    std::vector<std::string> names;
    std::vector<prf_seq> seqs;
};

// Copies [p, q) behind s without the bytes <= ' ' (newlines, a '\r' before them, blanks): eight bytes at a time
// where a word holds none of them, else byte by byte without a branch.
static inline void append_clean(prf_seq &s, const char *p, const char *q) {
    s.room((size_t)(q - p));
    char *out = s.p + s.n;
    const uint64_t ones = 0x0101010101010101ull;
    while (q - p >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        if (((w - ones * 0x21u) & ~w & (ones * 0x80u)) == 0) {  // no byte below 0x21
            memcpy(out, &w, 8);
            out += 8;
            p += 8;
        } else {
            for (int i = 0; i < 8; i++) {
                const char c = p[i];
                *out = c;
                out += (unsigned char)c > ' ';
            }
            p += 8;
        }
    }
    for (; p < q; p++) {
        *out = *p;
        out += (unsigned char)*p > ' ';
    }
    s.n = (size_t)(out - s.p);
}

extern "C" {

static inline void add_record(prf_fasta *fa, const char *h, const char *hend) {
    const char *a = h;  // name = header text up to the first white space
    while (a < hend && *a != ' ' && *a != '\t' && *a != '\r') a++;
    fa->names.emplace_back(h, (size_t)(a - h));
    fa->seqs.emplace_back();
}

// uncompressed file, mapped: records located first (a '>' at the start of a line), so that every sequence is
// reserved once at (nearly) its final size and each line is one bounded copy
static void parse_mapped(prf_fasta *fa, const char *base, size_t n) {
    const char *end = base + n;
    std::vector<const char *> heads;
    for (const char *p = base; p < end;) {
        const char *g = (const char *)memchr(p, '>', (size_t)(end - p));
        if (!g) break;
        if (g == base || g[-1] == '\n') heads.push_back(g);
        p = g + 1;
    }
    for (size_t i = 0; i < heads.size(); i++) {
        const char *h = heads[i] + 1, *rec_end = i + 1 < heads.size() ? heads[i + 1] : end;
        const char *nl = (const char *)memchr(h, '\n', (size_t)(rec_end - h));
        add_record(fa, h, nl ? nl : rec_end);
        if (!nl) continue;
        prf_seq &seq = fa->seqs.back();
        seq.reserve((size_t)(rec_end - nl));
        append_clean(seq, nl + 1, rec_end);
    }
}

int prf_fasta_open(const char *path, prf_fasta **out) {
    if (!path || !out) return prf_set_error(PRF_EINVAL, "prf_fasta_open: bad arguments");
    *out = nullptr;
    prf_fasta *fa = new (std::nothrow) prf_fasta();
    if (!fa) return prf_set_error(PRF_ENOMEM, "prf_fasta_open: out of memory");
    try {
        // plain text: map the file (no copy through zlib's buffers)
        const int fd = open(path, O_RDONLY);
        if (fd < 0) {
            delete fa;
            return prf_set_error(PRF_EINVAL, "prf_fasta_open: cannot open %s", path);
        }
        unsigned char magic[2] = {0, 0};
        struct stat st;
        const bool is_file = fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
        const bool gz = pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (is_file && !gz) {
            if (st.st_size > 0) {
                void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) {
                    (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                    parse_mapped(fa, (const char *)m, (size_t)st.st_size);
                    munmap(m, (size_t)st.st_size);
                    close(fd);
                    *out = fa;
                    return PRF_OK;
                }
            } else {
                close(fd);
                *out = fa;
                return PRF_OK;
            }
        }
        close(fd);
    } catch (const std::bad_alloc &) {
        delete fa;
        return prf_set_error(PRF_ENOMEM, "prf_fasta_open: out of memory reading %s", path);
    }
    // gzip (or not a regular file): stream through zlib
    gzFile f = gzopen(path, "rb");
    if (!f) {
        delete fa;
        return prf_set_error(PRF_EINVAL, "prf_fasta_open: cannot open %s", path);
    }
    gzbuffer(f, 1 << 20);
    try {
        std::vector<char> buf(1 << 22);
        bool in_header = false, at_line_start = true, have_record = false;
        std::string header;
        for (;;) {
            const int n = gzread(f, buf.data(), (unsigned)buf.size());
            if (n < 0) {
                int errnum = 0;
                const char *msg = gzerror(f, &errnum);
                std::string m = msg ? msg : "read error";
                gzclose(f);
                delete fa;
                return prf_set_error(PRF_EINVAL, "prf_fasta_open: %s: %s", path, m.c_str());
            }
            if (n == 0) break;
            const char *p = buf.data(), *end = p + n;
            while (p < end) {
                if (in_header) {
                    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
                    header.append(p, nl ? nl : end);
                    if (!nl) break;
                    add_record(fa, header.data(), header.data() + header.size());
                    have_record = true;
                    header.clear();
                    in_header = false;
                    at_line_start = true;
                    p = nl + 1;
                } else if (at_line_start && *p == '>') {
                    in_header = true;
                    p++;
                } else {
                    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
                    const char *stop = nl ? nl : end;
                    if (have_record) append_clean(fa->seqs.back(), p, stop);
                    at_line_start = nl != nullptr;
                    p = nl ? nl + 1 : end;
                }
            }
        }
        if (in_header) add_record(fa, header.data(), header.data() + header.size());  // header without a newline at the end of the file
    } catch (const std::bad_alloc &) {
        gzclose(f);
        delete fa;
        return prf_set_error(PRF_ENOMEM, "prf_fasta_open: out of memory reading %s", path);
    }
    gzclose(f);
    *out = fa;
    return PRF_OK;
}

// One contig only.  With a samtools-style index next to an uncompressed file (path + ".fai": name, length, byte offset,
// bases per line, bytes per line) the record is read by seeking (what pyfastx's own index gives the reference's
// fasta[chrom].seq, perfect_repeat_finder.py:130); otherwise the file is parsed and the other records dropped.
// *out holds one entry, or none if the name is absent.
int prf_fasta_open_contig(const char *path, const char *name, prf_fasta **out) {
    if (!path || !name || !out) return prf_set_error(PRF_EINVAL, "prf_fasta_open_contig: bad arguments");
    *out = nullptr;
    const std::string p(path), want(name);
    const bool gz = p.size() > 3 && p.compare(p.size() - 3, 3, ".gz") == 0;
    try {
        if (!gz) {
            if (FILE *fi = fopen((p + ".fai").c_str(), "rb")) {
                char line[4096];
                unsigned long long len = 0, off = 0, lb = 0, lw = 0;
                bool found = false;
                while (fgets(line, sizeof line, fi)) {
                    char *tab = strchr(line, '\t');
                    if (!tab) continue;
                    if ((size_t)(tab - line) == want.size() && memcmp(line, want.data(), want.size()) == 0 &&
                        sscanf(tab + 1, "%llu\t%llu\t%llu\t%llu", &len, &off, &lb, &lw) == 4) {
                        found = true;
                        break;
                    }
                }
                fclose(fi);
                if (found && lb > 0 && lw >= lb) {
                    if (FILE *f = fopen(path, "rb")) {
                        // the index is untrusted input: the span it names must lie inside the file, and the line in front of
                        // it must be the header of the record asked for (a stale index may point into another record of the
                        // same length); anything else -> the full parse below
                        struct stat sb;
                        bool ok = fstat(fileno(f), &sb) == 0 && off <= (unsigned long long)sb.st_size;
                        const unsigned long long room = ok ? (unsigned long long)sb.st_size - off : 0;
                        unsigned long long raw = len + (len / lb + 1) * (lw - lb);
                        if (lw - lb > 8 || len > room) ok = false;
                        if (raw > room) raw = room;
                        if (ok) {  // header check: bytes in front of `off` are ">name" + optional description + newline
                            const unsigned long long back = std::min<unsigned long long>(off, 4096);
                            std::string head((size_t)back, '\0');
                            ok = back > 0 && fseeko(f, (off_t)(off - back), SEEK_SET) == 0 && fread(&head[0], 1, (size_t)back, f) == back &&
                                 head.back() == '\n';
                            if (ok) {
                                size_t ls = head.rfind('\n', head.size() - 2);
                                ls = ls == std::string::npos ? (back == off ? 0 : std::string::npos) : ls + 1;
                                ok = ls != std::string::npos && head[ls] == '>' && head.compare(ls + 1, want.size(), want) == 0 &&
                                     (ls + 1 + want.size() >= head.size() || isspace((unsigned char)head[ls + 1 + want.size()]));
                            }
                        }
                        std::unique_ptr<prf_fasta> fa;
                        if (ok) {
                            std::string buf((size_t)raw, '\0');
                            ok = fseeko(f, (off_t)off, SEEK_SET) == 0;
                            const size_t got = ok ? fread(&buf[0], 1, (size_t)raw, f) : 0;
                            fa.reset(new prf_fasta());
                            fa->names.emplace_back(want);
                            fa->seqs.emplace_back();
                            prf_seq &seq = fa->seqs.back();
                            size_t stop = 0;  // a '>' inside the span: the index does not match the file
                            while (stop < got && buf[stop] != '>') stop++;
                            append_clean(seq, buf.data(), buf.data() + stop);
                            // exactly `len` bases, and the record ends there: the next byte that is no line end is the '>' of
                            // the next header or the end of the file.  An index whose length is too short would otherwise
                            // hand over a truncated record without a word.
                            ok = seq.size() == len;
                            if (ok && stop == got) {
                                int ch;
                                while ((ch = fgetc(f)) == '\n' || ch == '\r') {}
                                ok = ch == EOF || ch == '>';
                            }
                        }
                        fclose(f);
                        if (ok) {
                            *out = fa.release();
                            return PRF_OK;
                        }
                        // stale or damaged index: fall through to the full parse
                    }
                }
            }
        }
        prf_fasta *all = nullptr;
        const int rc = prf_fasta_open(path, &all);
        if (rc != PRF_OK) return rc;
        prf_fasta *fa = new prf_fasta();
        for (size_t i = 0; i < all->names.size(); i++)
            if (all->names[i] == want) {
                fa->names.emplace_back(std::move(all->names[i]));
                fa->seqs.emplace_back(std::move(all->seqs[i]));
                break;
            }
        delete all;
        *out = fa;
        return PRF_OK;
    } catch (const std::bad_alloc &) {
        return prf_set_error(PRF_ENOMEM, "prf_fasta_open_contig: out of memory reading %s", path);
    } catch (...) {  // nothing may cross the C boundary
        return prf_set_error(PRF_EINVAL, "prf_fasta_open_contig: unexpected failure reading %s", path);
    }
}

int prf_fasta_count(const prf_fasta *f) { return f ? (int)f->names.size() : 0; }

int prf_fasta_entry(const prf_fasta *f, int i, const char **name, const uint8_t **seq, uint64_t *len) {
    if (!f || i < 0 || i >= (int)f->names.size()) return prf_set_error(PRF_EINVAL, "prf_fasta_entry: index out of range");
    if (name) *name = f->names[i].c_str();
    if (seq) *seq = (const uint8_t *)f->seqs[i].data();
    if (len) *len = f->seqs[i].size();
    return PRF_OK;
}

void prf_fasta_close(prf_fasta *f) { delete f; }

static int write_rows(const char *path, int append, const char *header, const char *const *names, const prf_contig *contigs,
                      int n_contigs, const prf_hits *hits, uint64_t *n_written) {
    if (!path || !hits || (hits->n && (!contigs || !hits->rows))) return prf_set_error(PRF_EINVAL, "row writer: bad arguments");
    FILE *f = fopen(path, append ? "ab" : "wb");
    if (!f) return prf_set_error(PRF_EINVAL, "cannot open %s for writing", path);
    std::string out;
    out.reserve(1 << 22);
    if (header) out += header;
    char num[64];
    for (uint64_t i = 0; i < hits->n; i++) {
        const prf_hit &h = hits->rows[i];
        if ((int)h.contig >= n_contigs || h.start + h.k > contigs[h.contig].len) {
            fclose(f);
            return prf_set_error(PRF_EINVAL, "row %llu does not fit its contig", (unsigned long long)i);
        }
        if (names) {
            out += names[h.contig];
            out += '\t';
        }
        const int n = snprintf(num, sizeof num, "%llu\t%llu\t", (unsigned long long)h.start, (unsigned long long)h.end);
        out.append(num, (size_t)n);
        const uint8_t *m = contigs[h.contig].ascii + h.start;
        for (uint32_t j = 0; j < h.k; j++) {
            const uint8_t c = m[j];
            out += (char)((c >= 'a' && c <= 'z') ? c - 32 : c);  // str.upper() on ASCII (reference :33)
        }
        out += '\n';
        if (out.size() > (1u << 22) - 4096) {
            if (fwrite(out.data(), 1, out.size(), f) != out.size()) {
                fclose(f);
                return prf_set_error(PRF_EINVAL, "write to %s failed", path);
            }
            out.clear();
        }
    }
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    if (fclose(f) != 0 || !ok) return prf_set_error(PRF_EINVAL, "write to %s failed", path);
    if (n_written) *n_written = hits->n;
    return PRF_OK;
}

int prf_write_bed(const char *path, int append, const char *const *names, const prf_contig *contigs, int n_contigs,
                  const prf_hits *hits, uint64_t *n_written) {
    if (!names) return prf_set_error(PRF_EINVAL, "prf_write_bed: names missing");
    return write_rows(path, append, nullptr, names, contigs, n_contigs, hits, n_written);
}

int prf_write_tsv(const char *path, const prf_contig *contig, const prf_hits *hits, uint64_t *n_written) {
    return write_rows(path, 0, "start_0based\tend\tmotif\n", nullptr, contig, 1, hits, n_written);
}

}  // extern "C"
