// C++ host-side mirror of the reference package's public API over the C ABI of
// lzma_amd.h.  Same names, argument meaning and error behaviour as the Go package:
//
//   NewReader                                  lzma_decoder.go:351-363
//   NewWriterSizeLevel / Level / Size / NewWriter   lzma_encoder.go:1109-1150
//   BestSpeed / BestCompression / DefaultCompression  lzma_encoder.go:14-18
//   HeaderError / StreamError                  lzma_decoder.go:91-95
//   ArgumentValueError                         lzma_encoder.go:26-34
//
// The reference streams through a goroutine and a pipe; the bytes it produces depend
// only on the whole input, so the writer buffers what is written and encodes on Close(),
// and the reader decodes on the first Read().  Errors are exceptions where Go returns an
// error value; as in the reference, an encoder argument error surfaces on the first
// Write or on Close, not in the constructor.  EncodeChunks / DecodeStreams are the batch
// forms that keep the whole GPU busy.  Header-only; link with liblzma_amd.so.
#ifndef LZMA_AMD_HPP
#define LZMA_AMD_HPP

#include <cstdint>
#include <cstring>
#include <istream>
#include <iterator>
#include <memory>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "lzma_amd.h"

namespace lzma {

constexpr int BestSpeed = 1;
constexpr int BestCompression = 9;
constexpr int DefaultCompression = 5;

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct HeaderError : Error {  // headerError, lzma_decoder.go:95
    HeaderError() : Error("error in lzma header") {}
};
struct StreamError : Error {  // streamError, lzma_decoder.go:92
    StreamError() : Error("error in lzma encoded data stream") {}
};
struct UnexpectedEOF : Error {  // io.ErrUnexpectedEOF from the 13-byte header read
    UnexpectedEOF() : Error("unexpected EOF") {}
};
struct ArgumentValueError : Error {  // lzma_encoder.go:26-34
    std::string msg;
    long long val;
    ArgumentValueError(const std::string &m, long long v)
        : Error("illegal argument value error: " + m + " with value " + std::to_string(v)), msg(m), val(v) {}
};
struct DeviceError : Error {  // no counterpart: the GPU or the library failed
    int code;
    explicit DeviceError(int c) : Error(std::string("lzma_amd: ") + lzma_amd_last_error()), code(c) {}
};

// ---- batch forms ---------------------------------------------------------------

// Every chunk becomes an independent .lzma stream (size-known header unless p.eos).
inline std::vector<std::string> EncodeChunks(const std::vector<std::string> &chunks, const lzma_amd_params &p) {
    const size_t n = chunks.size();
    std::vector<std::string> res(n);
    if (n == 0) return res;
    std::vector<uint64_t> in_off(n + 1, 0), out_off(n + 1, 0), sizes(n, 0);
    std::vector<int32_t> status(n, 0);
    for (size_t i = 0; i < n; i++) {
        in_off[i + 1] = in_off[i] + chunks[i].size();
        out_off[i + 1] = out_off[i] + lzma_amd_encode_bound(chunks[i].size());
    }
    std::string in;
    in.reserve(in_off[n]);
    for (const auto &c : chunks) in += c;
    std::string out(out_off[n], '\0');
    const int rc = lzma_amd_encode_chunks((const uint8_t *)in.data(), in_off.data(), n, &p, (uint8_t *)&out[0],
                                          out_off.data(), sizes.data(), status.data());
    if (rc == LZMA_AMD_ERR_ARG) throw ArgumentValueError(lzma_amd_last_error(), 0);
    if (rc != LZMA_AMD_OK) throw DeviceError(rc);
    for (size_t i = 0; i < n; i++) {
        if (status[i] != LZMA_AMD_OK) throw DeviceError(status[i]);
        res[i] = out.substr(out_off[i], sizes[i]);
    }
    return res;
}

struct Decoded {
    std::string data;
    int status;  // LZMA_AMD_OK, _ERR_HEADER, _ERR_STREAM, _ERR_TRUNCATED, ...
};

// What a stream of n compressed bytes can expand to at most: the cheapest symbol is a
// 273-byte rep0 match coded in 14 binary decisions, none cheaper than log2(2048/2017) bits,
// i.e. fewer than 7100 output bytes per compressed byte; one stream produces at most
// 4 GiB - 512 bytes (lzma_amd.h).  Capacities taken from an untrusted header are clamped to
// this, so a forged size field cannot make a batch allocate terabytes.
inline uint64_t MaxOutput(uint64_t n) {
    const uint64_t lim = ((uint64_t)1 << 32) - 512, v = 64 + 7200 * n;
    return v < lim ? v : lim;
}

// Complete .lzma streams (13-byte header + payload).  Output capacity comes from the
// header's size field (clamped to MaxOutput); for streams that end with a marker instead,
// from `eos_cap` (bytes per stream), or 40x the compressed size if that is 0 -- and every
// such stream that reports LZMA_AMD_ERR_OUT_CAP is submitted again with eight times the
// room until it fits or reaches MaxOutput.
inline std::vector<Decoded> DecodeStreams(const std::vector<std::string> &streams, uint64_t eos_cap = 0) {
    const size_t n = streams.size();
    std::vector<Decoded> res(n);
    if (n == 0) return res;
    std::vector<uint64_t> cap(n);
    std::vector<char> grow(n, 0);
    std::vector<size_t> todo(n);
    for (size_t i = 0; i < n; i++) {
        todo[i] = i;
        const uint64_t lim = MaxOutput(streams[i].size());
        uint32_t lc, lp, pb, dict;
        int64_t unpack = -1;
        if (lzma_amd_parse_header((const uint8_t *)streams[i].data(), streams[i].size(), &lc, &lp, &pb, &dict,
                                  &unpack) == LZMA_AMD_OK && unpack >= 0) {
            cap[i] = (uint64_t)unpack < lim ? (uint64_t)unpack : lim;
        } else {
            cap[i] = eos_cap ? eos_cap : 64 + 40 * (uint64_t)streams[i].size();
            if (cap[i] > lim) cap[i] = lim;
            grow[i] = 1;
        }
    }
    while (!todo.empty()) {
        const size_t m = todo.size();
        std::vector<uint64_t> in_off(m + 1, 0), out_off(m + 1, 0), sizes(m, 0);
        std::vector<int32_t> status(m, 0);
        std::string in;
        for (size_t k = 0; k < m; k++) {
            in_off[k + 1] = in_off[k] + streams[todo[k]].size();
            out_off[k + 1] = out_off[k] + cap[todo[k]];
            in += streams[todo[k]];
        }
        std::string out(out_off[m] ? out_off[m] : 1, '\0');
        const int rc = lzma_amd_decode_batch((const uint8_t *)in.data(), in_off.data(), m, (uint8_t *)&out[0],
                                             out_off.data(), sizes.data(), status.data());
        if (rc != LZMA_AMD_OK) throw DeviceError(rc);
        std::vector<size_t> again;
        for (size_t k = 0; k < m; k++) {
            const size_t i = todo[k];
            res[i].status = status[k];
            res[i].data = out.substr(out_off[k], sizes[k]);
            const uint64_t lim = MaxOutput(streams[i].size());
            if (status[k] == LZMA_AMD_ERR_OUT_CAP && grow[i] && cap[i] < lim) {
                cap[i] = 8 * cap[i] < lim ? 8 * cap[i] : lim;
                again.push_back(i);
            }
        }
        todo.swap(again);
    }
    return res;
}

// ---- the reference's constructors over the batch forms ----------------------------

class Writer {  // io.WriteCloser
  public:
    Writer(std::ostream &w, int64_t size, int level) : w_(w), size_(size), level_(level) {
        // reported by the first Write or by Close, as the encoder goroutine does (lzma_encoder.go:997-1010)
        if (level < 1 || level > 9) err_.reset(new ArgumentValueError("level out of range", level));
        else if (size < -1) err_.reset(new ArgumentValueError("illegal size", size));
    }
    size_t Write(const void *p, size_t n) {
        if (err_) throw *err_;
        buf_.append((const char *)p, n);
        return n;
    }
    size_t Write(const std::string &s) { return Write(s.data(), s.size()); }
    void Close() {  // blocks until the stream is written, like syncPipeWriter.Close (lzma_encoder.go:72-83)
        if (closed_) return;
        closed_ = true;
        if (err_) throw *err_;
        lzma_amd_params p;
        if (lzma_amd_params_from_level(level_, size_ == -1, &p) != LZMA_AMD_OK)
            throw ArgumentValueError("level out of range", level_);
        std::string out = EncodeChunks({buf_}, p)[0];
        if (size_ != -1)  // the caller's size goes into the header verbatim (lzma_encoder.go:1012-1031)
            for (int i = 0; i < 8; i++) out[5 + i] = (char)(((uint64_t)size_ >> (8 * i)) & 0xFF);
        w_.write(out.data(), (std::streamsize)out.size());
        w_.flush();
    }

  private:
    std::ostream &w_;
    int64_t size_;
    int level_;
    std::string buf_;
    std::unique_ptr<ArgumentValueError> err_;
    bool closed_ = false;
};

inline std::unique_ptr<Writer> NewWriterSizeLevel(std::ostream &w, int64_t size, int level) {
    return std::unique_ptr<Writer>(new Writer(w, size, level));
}
inline std::unique_ptr<Writer> NewWriterLevel(std::ostream &w, int level) { return NewWriterSizeLevel(w, -1, level); }
inline std::unique_ptr<Writer> NewWriterSize(std::ostream &w, int64_t size) {
    return NewWriterSizeLevel(w, size, DefaultCompression);
}
inline std::unique_ptr<Writer> NewWriter(std::ostream &w) { return NewWriterSizeLevel(w, -1, DefaultCompression); }

class Reader {  // io.ReadCloser
  public:
    explicit Reader(std::istream &r) : r_(r) {}
    // Up to n bytes; 0 at the end of the stream.  Header and data errors are thrown by the
    // first call (the reference's decoder goroutine reports them through the pipe).
    size_t Read(void *p, size_t n) {
        if (!filled_) fill();
        const size_t k = n < buf_.size() - pos_ ? n : buf_.size() - pos_;
        std::memcpy(p, buf_.data() + pos_, k);
        pos_ += k;
        return k;
    }
    std::string ReadAll() {
        if (!filled_) fill();
        std::string s = buf_.substr(pos_);
        pos_ = buf_.size();
        return s;
    }
    void Close() {}

  private:
    void fill() {
        filled_ = true;
        const std::string raw((std::istreambuf_iterator<char>(r_)), std::istreambuf_iterator<char>());
        Decoded d = DecodeStreams({raw})[0];
        if (d.status == LZMA_AMD_ERR_HEADER) throw HeaderError();
        if (d.status == LZMA_AMD_ERR_STREAM) throw StreamError();
        if (d.status == LZMA_AMD_ERR_TRUNCATED) {
            // the reference surfaces a plain io.EOF when the payload ends early (range_coder.go:82-85),
            // which a copy loop treats as a clean end; a short header is io.ErrUnexpectedEOF
            if (raw.size() < 13) throw UnexpectedEOF();
        } else if (d.status != LZMA_AMD_OK) {
            throw DeviceError(d.status);
        }
        buf_ = std::move(d.data);
    }
    std::istream &r_;
    std::string buf_;
    size_t pos_ = 0;
    bool filled_ = false;
};

inline std::unique_ptr<Reader> NewReader(std::istream &r) { return std::unique_ptr<Reader>(new Reader(r)); }

}  // namespace lzma

#endif
