// Run-time compilation of shape-specialised kernels.  The kernels built ahead of time
// (qe_static.hip) cover a fixed list of launch shapes; for any other shape the library generates
// the same Spec from the launch it is about to make, compiles qe_static.h's kernel template for
// it with the HIP run-time compiler, and uses the result for every later launch of that shape
// (in this process, and through an on-disk cache in later ones).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "qe_specgen.h"

namespace qe {

enum JitMode : int {
  JIT_OFF = 0,
  JIT_ASYNC = 1,  // compile on a background thread; launches use the interpreter until it is ready
  JIT_SYNC = 2    // compile on first use and wait (tests, prepared workloads)
};

// JIT_AGGREGATE_SPILL: the high-cardinality mode's first pass (qe_aggregate_spill_static) of an aggregate shape; its
// SpecFields carry one more signature word than the plain aggregate's (spec_fields_spill), so the two never share an entry.
// JIT_AGGREGATE_DIRECT: that mode's first pass that partitions its rows itself (qe_aggregate_direct_static; spec_fields_direct).
enum JitKind : int { JIT_PROJECT = 0, JIT_FILTER = 1, JIT_AGGREGATE = 2, JIT_AGGREGATE_SPILL = 3, JIT_AGGREGATE_DIRECT = 4 };

// Launches the run-time compiled kernel for this shape if there is one (or, in JIT_SYNC mode,
// after compiling it).  Returns false when the caller should use the interpreter instead: shape
// not specialisable, compilation still running or failed.
bool jit_launch(JitKind kind, const SpecFields &fields, const void *args, int mode, int device, unsigned grid, unsigned block,
                size_t lds_bytes, hipStream_t stream, hipError_t *err, const char **kernel_name);

// Is the kernel for this shape ready to launch?  In JIT_SYNC mode it is compiled now if need be; otherwise this only
// looks - and, with `start_if_absent`, starts the background compilation of a shape it has not seen (for kernels
// that no launch would otherwise ask for: a caller that picks its launch geometry by what is ready).
bool jit_ready(JitKind kind, const SpecFields &fields, int mode, bool start_if_absent = false);

// Compiles the kernel for `fields` without touching a device; returns the code object's size.
// Throws qe::Failure when the compiler rejects the source (the log is the message).
size_t jit_compile_only(JitKind kind, const SpecFields &fields);

int jit_default_mode();  // from the environment variable QE_JIT = off | async | sync (default async)

}  // namespace qe
