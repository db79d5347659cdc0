#include "k2w_file.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>

#include "errors.h"

namespace k2hip {

namespace {
// header cursor: every read is checked against the end of the header region
struct Cursor {
    const uint8_t* p;
    size_t q, lim;
    const char* path;
    void need(uint64_t n, const char* what) const {
        if (n > lim - q) failf(K2HIP_ERR_IO, "%s: truncated or corrupt header (%s at byte %zu)", path, what, q);
    }
    template <typename T>
    T rd(const char* what) {
        need(sizeof(T), what);
        T v;
        memcpy(&v, p + q, sizeof(T));
        q += sizeof(T);
        return v;
    }
    std::string str(uint64_t n, const char* what) {
        need(n, what);
        std::string s(reinterpret_cast<const char*>(p + q), (size_t)n);
        q += (size_t)n;
        return s;
    }
};
}  // namespace

K2wFile::K2wFile(const std::string& path) {
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) failf(K2HIP_ERR_IO, "cannot open weights file %s", path.c_str());
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 24) {
        close(fd);
        failf(K2HIP_ERR_IO, "%s: not a K2W1 container (not a regular file of at least 24 bytes)", path.c_str());
    }
    size_ = (size_t)st.st_size;
    void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) failf(K2HIP_ERR_IO, "mmap failed for %s", path.c_str());
    base_ = static_cast<const uint8_t*>(m);
    try {
        if (memcmp(base_, "K2W1", 4) != 0) failf(K2HIP_ERR_IO, "%s: not a K2W1 container", path.c_str());
        Cursor c{base_, 4, 24, path.c_str()};
        const uint32_t version = c.rd<uint32_t>("version"), n_meta = c.rd<uint32_t>("n_meta"), n_tensors = c.rd<uint32_t>("n_tensors");
        data_off_ = c.rd<uint64_t>("data_offset");
        if (version != 1) failf(K2HIP_ERR_IO, "%s: unsupported K2W version %u", path.c_str(), version);
        if (data_off_ < 24 || data_off_ > size_ || data_off_ % 4 != 0)
            failf(K2HIP_ERR_IO, "%s: data offset %llu outside the file (%zu bytes)", path.c_str(), (unsigned long long)data_off_, size_);
        c.lim = (size_t)data_off_;  // metadata and the tensor table live in front of the data region
        // each record needs at least 8 / 60 bytes: an absurd count fails here instead of in a long loop
        if ((uint64_t)n_meta * 8 > c.lim || (uint64_t)n_tensors * 60 > c.lim)
            failf(K2HIP_ERR_IO, "%s: header claims %u metadata entries and %u tensors in %zu bytes", path.c_str(), n_meta, n_tensors, c.lim);
        for (uint32_t i = 0; i < n_meta; i++) {
            const uint32_t kl = c.rd<uint32_t>("metadata key length"), vl = c.rd<uint32_t>("metadata value length");
            std::string k = c.str(kl, "metadata key");
            meta[k] = c.str(vl, "metadata value");
        }
        const uint64_t db = size_ - data_off_;
        tensors.reserve(n_tensors);
        for (uint32_t i = 0; i < n_tensors; i++) {
            K2wTensorRec r;
            const uint32_t nl = c.rd<uint32_t>("tensor name length");
            r.name = c.str(nl, "tensor name");
            r.dtype = c.rd<uint32_t>("dtype");
            const uint32_t ndim = c.rd<uint32_t>("ndim");
            if (ndim > 4) failf(K2HIP_ERR_IO, "%s: tensor %s has %u dimensions", path.c_str(), r.name.c_str(), ndim);
            r.ndim = (int)ndim;
            uint64_t numel = 1;
            for (int k = 0; k < 4; k++) {
                const uint64_t d = c.rd<uint64_t>("dims");
                if (d == 0 || d > ((uint64_t)1 << 40) || numel > (((uint64_t)1 << 48) / d))
                    failf(K2HIP_ERR_IO, "%s: tensor %s has an impossible shape", path.c_str(), r.name.c_str());
                if (k >= r.ndim && d != 1) failf(K2HIP_ERR_IO, "%s: tensor %s: dims beyond ndim must be 1", path.c_str(), r.name.c_str());
                numel *= d;
                r.dims[k] = (int64_t)d;
            }
            r.off = c.rd<uint64_t>("offset");
            r.nbytes = c.rd<uint64_t>("nbytes");
            if (r.dtype > 1) failf(K2HIP_ERR_IO, "%s: tensor %s has unknown dtype %u", path.c_str(), r.name.c_str(), r.dtype);
            const uint64_t esz = r.dtype == 0 ? 4 : 8;
            if (r.nbytes != numel * esz)
                failf(K2HIP_ERR_IO, "%s: tensor %s: %llu bytes for %llu elements", path.c_str(), r.name.c_str(),
                      (unsigned long long)r.nbytes, (unsigned long long)numel);
            if (r.off % esz != 0 || r.off > db || r.nbytes > db - r.off)  // overflow-safe range check
                failf(K2HIP_ERR_IO, "%s: tensor %s lies outside the file", path.c_str(), r.name.c_str());
            tensors.push_back(std::move(r));
        }
    } catch (...) {
        munmap(const_cast<uint8_t*>(base_), size_);
        base_ = nullptr;
        throw;
    }
}

K2wFile::~K2wFile() {
    if (base_) munmap(const_cast<uint8_t*>(base_), size_);
}

}  // namespace k2hip
