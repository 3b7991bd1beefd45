// libzling_utils.h -- the I/O and progress interfaces of baidu::zling, kept source-compatible
// with the reference (src/libzling_utils.h:48-119) so existing callers compile unchanged:
//
//   Inputter       pull source   GetData / IsEnd / IsErr  (+ GetChar, GetUInt32 big-endian)
//   Outputter      push sink     PutData / IsErr          (+ PutChar, PutUInt32 big-endian)
//   ActionHandler  callbacks     OnInit / OnProcess(raw block, size) / OnDone
//   FileInputter / FileOutputter   stdio-backed implementations with byte counters
//
// Semantics preserved from the reference: short reads/writes are allowed (the drivers loop),
// no virtual destructors, callbacks run on the caller's thread in stream order.
#ifndef LIBZLING_AMD_UTILS_H
#define LIBZLING_AMD_UTILS_H

#include "libzling_inc.h"

namespace baidu {
namespace zling {

struct Inputter {
    virtual size_t GetData(unsigned char* buf, size_t len) = 0;   // returns bytes delivered (may be short)
    virtual bool IsEnd() = 0;
    virtual bool IsErr() = 0;

    int GetChar();            // one byte (undefined at end of input, like the reference)
    uint32_t GetUInt32();     // four bytes, most significant first
};

struct Outputter {
    virtual size_t PutData(unsigned char* buf, size_t len) = 0;   // returns bytes accepted (may be short)
    virtual bool IsErr() = 0;

    int PutChar(int v);
    uint32_t PutUInt32(uint32_t v);   // most significant byte first
};

struct ActionHandler {
    virtual void OnInit() {}
    virtual void OnDone() {}
    virtual void OnProcess(unsigned char* orig_data, size_t orig_size) { (void)orig_data; (void)orig_size; }

    void SetInputterOutputter(Inputter* inputter, Outputter* outputter, bool is_encode) {
        encode_side_ = is_encode;
        in_ = inputter;
        out_ = outputter;
    }
    bool IsEncode() { return encode_side_; }
    Inputter* GetInputter() { return in_; }
    Outputter* GetOutputter() { return out_; }

private:
    bool encode_side_;
    Inputter* in_;
    Outputter* out_;
};

struct FileInputter : public Inputter {
    FileInputter(FILE* fp) : fp_(fp), consumed_(0) {}
    size_t GetData(unsigned char* buf, size_t len);
    bool IsEnd();
    bool IsErr();
    size_t GetInputSize();

private:
    FILE* fp_;
    size_t consumed_;
};

struct FileOutputter : public Outputter {
    FileOutputter(FILE* fp) : fp_(fp), produced_(0) {}
    size_t PutData(unsigned char* buf, size_t len);
    bool IsErr();
    size_t GetOutputSize();

private:
    FILE* fp_;
    size_t produced_;
};

}  // namespace zling
}  // namespace baidu
#endif
