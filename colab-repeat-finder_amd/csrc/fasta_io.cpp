// fasta_io.cpp -- host-side FASTA reader and BED/TSV writers of libprf (no GPU code).
//
// The two data formats either side of the hot path:
//  * prf_fasta_*: what the reference gets from pyfastx.Fasta (reference perfect_repeat_finder.py:117,
//    :130, :136-143): entries in file order, name = header up to the first white space, sequence = the
//    record's lines joined, case preserved.  Plain or gzip-compressed text (zlib reads both).
//  * prf_write_bed / prf_write_tsv: the reference's output lines, "chrom\tstart\tend\tmotif\n"
//    (:148-149) and "start_0based\tend\tmotif" + rows (:166-170), motif = seq.upper()[start:start+k].
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/prf.h"

int prf_set_error(int code, const char *fmt, ...);  // api.cpp

struct prf_fasta {
    std::vector<std::string> names;
    std::vector<std::string> seqs;
};

extern "C" {

int prf_fasta_open(const char *path, prf_fasta **out) {
    if (!path || !out) return prf_set_error(PRF_EINVAL, "prf_fasta_open: bad arguments");
    *out = nullptr;
    gzFile f = gzopen(path, "rb");
    if (!f) return prf_set_error(PRF_EINVAL, "prf_fasta_open: cannot open %s", path);
    gzbuffer(f, 1 << 20);
    prf_fasta *fa = new (std::nothrow) prf_fasta();
    if (!fa) {
        gzclose(f);
        return prf_set_error(PRF_ENOMEM, "prf_fasta_open: out of memory");
    }
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
                    // name = text after '>' up to the first white space
                    size_t a = 0;
                    while (a < header.size() && header[a] != ' ' && header[a] != '\t' && header[a] != '\r') a++;
                    fa->names.emplace_back(header.substr(0, a));
                    fa->seqs.emplace_back();
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
                    if (have_record) {
                        const char *q = stop;
                        while (q > p && (unsigned char)q[-1] <= ' ') q--;  // trailing \r / blanks of the line
                        std::string &s = fa->seqs.back();
                        // inner white space (rare) is dropped as well
                        const char *r = p;
                        while (r < q) {
                            const char *w = r;
                            while (w < q && (unsigned char)*w > ' ') w++;
                            s.append(r, w);
                            r = w;
                            while (r < q && (unsigned char)*r <= ' ') r++;
                        }
                    }
                    at_line_start = nl != nullptr;
                    p = nl ? nl + 1 : end;
                }
            }
        }
        if (in_header) {  // header without a newline at the end of the file
            size_t a = 0;
            while (a < header.size() && header[a] != ' ' && header[a] != '\t' && header[a] != '\r') a++;
            fa->names.emplace_back(header.substr(0, a));
            fa->seqs.emplace_back();
        }
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
                        const unsigned long long raw = len + (len / lb + 1) * (lw - lb);
                        std::string buf(raw, '\0');
                        bool ok = fseeko(f, (off_t)off, SEEK_SET) == 0;
                        const size_t got = ok ? fread(&buf[0], 1, raw, f) : 0;
                        fclose(f);
                        prf_fasta *fa = new prf_fasta();
                        fa->names.emplace_back(want);
                        fa->seqs.emplace_back();
                        std::string &seq = fa->seqs.back();
                        seq.reserve(len);
                        for (size_t i = 0; i < got && seq.size() < len; i++) {
                            const unsigned char ch = (unsigned char)buf[i];
                            if (ch == '>') break;  // the index does not match the file
                            if (ch > ' ') seq += (char)ch;
                        }
                        if (seq.size() == len) {
                            *out = fa;
                            return PRF_OK;
                        }
                        delete fa;  // stale index: fall through to the full parse
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
