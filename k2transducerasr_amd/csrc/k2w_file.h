// Bounds-checked reader of the .k2w weight container (layout: k2transducerasr_amd/k2w.py).
// The reference reads ONNX files through ONNXRuntime and turns every failure into an exception
// (OfflineProjOfTransducer.cs:87-90); here a truncated, corrupted or mismatched file must come back as K2HIP_ERR_IO
// from k2hip_model_create, never as a crash.  Pure host code (no HIP): built and fuzzed under ASan/UBSan on the CPU.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace k2hip {

struct K2wTensorRec {
    std::string name;
    uint32_t dtype = 0;     // 0 = f32, 1 = i64 (kept in the table, never read by the engine)
    int ndim = 0;
    int64_t dims[4] = {1, 1, 1, 1};
    uint64_t off = 0;       // relative to the data region, 4-byte aligned, range-checked
    uint64_t nbytes = 0;    // == numel * sizeof(dtype), checked
};

class K2wFile {
  public:
    explicit K2wFile(const std::string& path);  // throws Error(K2HIP_ERR_IO) on anything inconsistent
    ~K2wFile();
    K2wFile(const K2wFile&) = delete;
    K2wFile& operator=(const K2wFile&) = delete;

    const uint8_t* data() const { return base_ + data_off_; }  // start of the data region
    size_t data_bytes() const { return size_ - data_off_; }
    std::map<std::string, std::string> meta;
    std::vector<K2wTensorRec> tensors;

  private:
    const uint8_t* base_ = nullptr;
    size_t size_ = 0;
    uint64_t data_off_ = 0;
};

}  // namespace k2hip
