// ref_binaryio_driver.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref).
//
// Own code.  A C-callable driver around the reference's *unmodified* on-disk codec,
// compiled by oracle/Makefile straight from /root/reference/src/binaryio.cpp (the only
// file of the query path that builds without absent third-party headers).  It replays
// an operation script through the real BitWriter / BitReader so that tests can pin
//   * cammiq_amd/synth.py's writer  (byte-identical files), and
//   * oracle/cammiq_oracle.c's and the product's readers (same values read back)
// against the reference itself.  Outputs live only in oracle/_ref/ (git-ignored).
//
// op codes: 0 bit(v)  1 bits(count=arg,v)  2 u16  3 u32  4 u64  5 flush64 (write only)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include "binaryio.hpp"   // -I/root/reference/src

extern "C" {

int ref_write_ops(const char *path, const uint8_t *op, const uint8_t *arg, const uint64_t *val, uint64_t n)
{
    BitWriter w;
    w.openFile(std::string(path));
    for (uint64_t i = 0; i < n; i++) {
        switch (op[i]) {
        case 0: w.writeBit(val[i] != 0); break;
        case 1: w.writeBits((int)arg[i], (uint32_t)val[i]); break;
        case 2: w.writeBits16((uint32_t)val[i]); break;
        case 3: w.writeBits32((uint32_t)val[i]); break;
        case 4: w.writeBits64(val[i]); break;
        case 5: w.flush64(); break;
        default: return -1;
        }
    }
    w.closeFile();
    return 0;
}

int ref_read_ops(const char *path, const uint8_t *op, const uint8_t *arg, uint64_t *val, uint64_t n)
{
    BitReader r;
    r.openFile(std::string(path));
    for (uint64_t i = 0; i < n; i++) {
        switch (op[i]) {
        case 0: val[i] = r.readBit(); break;
        case 1: val[i] = r.readBits((int)arg[i]); break;
        case 2: val[i] = r.readBits16(); break;
        case 3: val[i] = r.readBits32(); break;
        case 4: val[i] = r.readBits64(); break;
        default: return -1;
        }
    }
    r.closeFile();
    return 0;
}

}  // extern "C"
