// loader.h — the safetensors reader shared by loader.cpp (checkpoint ingestion) and dit.cpp (pre-quantized
// checkpoints): mmap + header parse, no copy.
#pragma once
#include <stdint.h>
#include <map>
#include <string>
#include <vector>

struct TensorInfo {
    std::string dtype;
    std::vector<int64_t> shape;
    uint64_t begin = 0, end = 0;   // byte range relative to the data section
};

struct MappedFile {
    int fd = -1;
    uint8_t* base = nullptr;
    size_t size = 0;
    ~MappedFile();
};

// Maps `path`, fills `out` (every tensor checked to lie inside the file) and `*data` (start of the data section);
// optionally the flat string metadata and the header length.  FLUX2MI_OK or an error with flux2mi_last_error set.
int parse_safetensors(const char* path, MappedFile& f, std::map<std::string, TensorInfo>& out, const uint8_t** data,
                      std::map<std::string, std::string>* metadata = nullptr, uint64_t* header_len = nullptr);
