// Native FASTA / FASTQ (plain or gzip) -> FlatFile, and per-record sequence lengths: the host side of SURVEY.md section 8
// row f-1 (/root/reference/src/fxstats.cpp:12-23 `getlens`, :33-64 `FlatFile::make`, :202-219 `getstats`).
//
// The reference parses with klib's kseq over zlib and keeps EVERY sequence in a std::vector<std::string> before it writes
// the file (fxstats.cpp:44-50,59-61).  This reader restates the same record grammar as a small state machine over gzread
// blocks (own code; the behaviour it reproduces is listed below and checked against the compiled reference on random
// inputs in tests/test_flatfile.py) and STREAMS: sequence bytes go to a temporary file as they are parsed, only the
// offsets (8 bytes per record) stay in memory, and the FlatFile is assembled at the end -- a 1M-read FASTQ needs ~8 MB.
//
// Record grammar (kseq.h:178-216 as vendored by the reference), byte for byte:
//   * outside a record, bytes are skipped up to the next '>' or '@' -- anywhere, not only at a line start;
//   * the name runs to the first whitespace byte; unless that byte is '\n' the rest of the line is a comment;
//   * the sequence is every following line whose FIRST byte is not '>', '+' or '@': empty lines are skipped, the line is
//     appended without its '\n', and after each appended line ONE trailing '\r' is dropped if the sequence is longer than
//     one byte (so "\r\n" files work; interior bytes, blanks included, are kept as they are);
//   * a line starting with '>' or '@' ends a FASTA record and opens the next one;
//   * a line starting with '+' makes it FASTQ: the rest of that line is skipped, then quality lines are appended (same
//     '\r' rule) until there are at least as many quality bytes as sequence bytes; the record counts only if both lengths
//     are EQUAL -- otherwise, or if the file ends before the quality, reading STOPS there (kseq_read returns < 0 and the
//     reference's `while (kseq_read(ks) >= 0)` loops end), keeping the records before it;
//   * after a FASTQ record the reader is outside a record again.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "bsq.h"
#include "bsq_internal.h"

namespace {

class Stream {  // byte stream over gzread (plain files pass through)
  public:
    explicit Stream(gzFile f) : f_(f), buf_(size_t(1) << 18) {}
    int getc() {
        if (begin_ >= end_ && !fill()) return -1;
        return static_cast<unsigned char>(buf_[size_t(begin_++)]);
    }
    // Appends the bytes up to (not including) the next '\n' to out and consumes the '\n'.  Returns false if nothing at all
    // could be read (end of file before any byte).
    bool rest_of_line(std::string *out, bool *any) {
        bool got = false;
        for (;;) {
            if (begin_ >= end_ && !fill()) break;
            got = true;
            const char *p = buf_.data() + begin_;
            const char *nl = static_cast<const char *>(std::memchr(p, '\n', size_t(end_ - begin_)));
            const int n = nl ? int(nl - p) : end_ - begin_;
            if (out) out->append(p, size_t(n));
            begin_ += n + (nl ? 1 : 0);
            if (nl) break;
        }
        if (any) *any = got;
        return got;
    }
    bool error() const { return err_; }

  private:
    bool fill() {
        if (eof_) return false;
        const int n = gzread(f_, buf_.data(), unsigned(buf_.size()));
        if (n <= 0) {
            eof_ = true;
            err_ = n < 0;
            return false;
        }
        begin_ = 0;
        end_ = n;
        return true;
    }
    gzFile f_;
    std::vector<char> buf_;
    int begin_ = 0, end_ = 0;
    bool eof_ = false, err_ = false;
};

inline bool is_space(int c) { return c == ' ' || (c >= '\t' && c <= '\r'); }  // isspace in the C locale

inline void drop_cr(std::string *s) {
    if (s->size() > 1 && s->back() == '\r') s->pop_back();
}

// Calls on_record(seq) for every record the reference's kseq loop would yield.  Returns false on a read error.
template <typename F>
bool for_each_record(gzFile f, F on_record) {
    Stream in(f);
    int last = 0;  // the header byte that the previous record's scan already consumed, or 0
    std::string seq, qual;
    for (;;) {
        int c;
        if (last == 0) {
            while ((c = in.getc()) >= 0 && c != '>' && c != '@') {
            }
            if (c < 0) break;
            last = c;
        }
        // name (to the first whitespace byte), comment (rest of the line)
        bool any_name = false;
        while ((c = in.getc()) >= 0) {
            any_name = true;
            if (is_space(c)) break;
        }
        if (!any_name) break;  // end of file right behind the header byte
        if (c >= 0 && c != '\n') in.rest_of_line(nullptr, nullptr);
        seq.clear();
        while ((c = in.getc()) >= 0 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            seq.push_back(char(c));
            in.rest_of_line(&seq, nullptr);
            drop_cr(&seq);
        }
        if (c == '>' || c == '@') last = c;
        if (c != '+') {  // FASTA record (or the end of the file)
            on_record(seq);
            if (c < 0) {
                // (the reference's next call finds no name and stops)
                break;
            }
            continue;
        }
        while ((c = in.getc()) >= 0 && c != '\n') {
        }
        if (c < 0) break;  // no quality string: the record does not count, reading stops
        qual.clear();
        for (;;) {
            bool any = false;
            in.rest_of_line(&qual, &any);
            if (!any) break;
            drop_cr(&qual);
            if (qual.size() >= seq.size()) break;
        }
        last = 0;
        if (qual.size() != seq.size()) break;  // truncated / over-long quality: reading stops here
        on_record(seq);
    }
    return !in.error();
}

struct GzCloser {
    gzFile f;
    ~GzCloser() {
        if (f) gzclose(f);
    }
};
struct FileCloser {
    std::FILE *f;
    ~FileCloser() {
        if (f) std::fclose(f);
    }
};

}  // namespace

extern "C" {

bsq_status bsq_fastx_lengths(const char *path, uint64_t *lens, int64_t capacity, int64_t *nrecords) {
    if (!path || !nrecords || capacity < 0 || (capacity > 0 && !lens))
        return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bsq_fastx_lengths: null pointer");
    GzCloser in{gzopen(path, "r")};
    if (!in.f) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, (std::string(path) + " failed to open").c_str());
    (void)gzbuffer(in.f, 1u << 18);
    int64_t n = 0;
    const bool ok = for_each_record(in.f, [&](const std::string &s) {
        if (n < capacity) lens[n] = s.size();
        ++n;
    });
    *nrecords = n;
    if (!ok) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, (std::string(path) + ": read error").c_str());
    return BSQ_OK;
}

bsq_status bsq_fastx_to_flatfile(const char *inpath, const char *outpath, int64_t *nseqs, int64_t *max_seq_len) {
    if (!inpath || !outpath || !*outpath) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bsq_fastx_to_flatfile: null path");
    GzCloser in{gzopen(inpath, "r")};
    if (!in.f) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, (std::string(inpath) + " failed to open").c_str());
    (void)gzbuffer(in.f, 1u << 18);
    const std::string tmp = std::string(outpath) + ".seq.tmp";
    FileCloser body{std::fopen(tmp.c_str(), "wb+")};
    if (!body.f) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, (tmp + " could not be opened for writing").c_str());
    std::vector<uint64_t> offsets{0};
    uint64_t longest = 0;
    bool too_long = false, write_failed = false;
    const bool ok = for_each_record(in.f, [&](const std::string &s) {
        if (s.size() > 0xFFFFFFFFull) too_long = true;  // fxstats.cpp:45
        if (s.size() > longest) longest = s.size();
        offsets.push_back(offsets.back() + s.size());
        if (!s.empty() && std::fwrite(s.data(), 1, s.size(), body.f) != s.size()) write_failed = true;
    });
    auto fail = [&](bsq_status st, const std::string &msg) {
        std::remove(tmp.c_str());
        return bsq_internal::set_error(st, msg.c_str());
    };
    if (!ok) return fail(BSQ_ERR_INVALID_ARG, std::string(inpath) + ": read error");
    if (too_long) return fail(BSQ_ERR_INVALID_ARG, "Cannot handle sequences longer than 2^32 - 1");
    if (write_failed || std::fflush(body.f) != 0) return fail(BSQ_ERR_INVALID_ARG, tmp + ": write error");
    // assemble: uint64 nseqs | uint64 offsets[nseqs + 1] | the bytes (fxstats.cpp:51-61)
    FileCloser out{std::fopen(outpath, "wb")};
    if (!out.f) return fail(BSQ_ERR_INVALID_ARG, std::string(outpath) + " could not be opened for writing");
    const uint64_t n = offsets.size() - 1;
    bool good = std::fwrite(&n, sizeof(n), 1, out.f) == 1 && std::fwrite(offsets.data(), sizeof(uint64_t), offsets.size(), out.f) == offsets.size();
    std::rewind(body.f);
    std::vector<char> block(size_t(1) << 20);
    for (size_t got; good && (got = std::fread(block.data(), 1, block.size(), body.f)) > 0;) good = std::fwrite(block.data(), 1, got, out.f) == got;
    good = good && std::ferror(body.f) == 0 && std::fflush(out.f) == 0;
    std::fclose(body.f);
    body.f = nullptr;
    std::remove(tmp.c_str());
    if (!good) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, (std::string(outpath) + ": write error").c_str());
    if (nseqs) *nseqs = int64_t(n);
    if (max_seq_len) *max_seq_len = int64_t(longest);
    return BSQ_OK;
}

}  // extern "C"
