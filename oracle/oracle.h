// ORACLE — TEST INFRASTRUCTURE ONLY.  C API of the CPU restatement, loaded via
// ctypes by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
// The product (structure-slam-pointline_amd/) never includes or links this.
#pragma once
#include <cstdint>
#include <cstddef>
