// Flat description ("signature") of one lowered launch: the interpreter code, the column-cache
// layout and the sink descriptors, without any address or literal value.  Two launches with the
// same signature differ only in pointers, literals and row counts, so a kernel specialised at
// build time for a signature can serve every such launch.
#pragma once
#include <stdint.h>

#include <vector>

#include "qe_vm.h"

namespace qe {

enum SigKind : uint32_t { SIG_PROJECT = 0x5101, SIG_FILTER = 0x5102, SIG_AGGREGATE = 0x5103 };

inline int program_words_used(const VmProgram &P) {
  int last = -1;
  for (int i = 0; i < VM_MAX_INSN; ++i)
    if (P.insn[i] != 0) last = i;
  return last + 2 > VM_MAX_INSN ? VM_MAX_INSN : last + 2;  // include the closing OP_END
}

inline void sig_program(const VmProgram &P, std::vector<uint32_t> &s) {
  int n = program_words_used(P);
  s.push_back((uint32_t)n);
  for (int i = 0; i < n; ++i) s.push_back(P.insn[i]);
  s.push_back(P.cache_n);
  for (uint32_t i = 0; i < P.cache_n; ++i) s.push_back(P.cache_kind[i] | (P.cols[P.cache_col[i]].validity ? 1u << 8 : 0u));
}

inline void sig_outs(const VmOut *outs, int n, std::vector<uint32_t> &s) {
  s.push_back((uint32_t)n);
  for (int i = 0; i < n; ++i) s.push_back(outs[i].pc | ((uint32_t)outs[i].type << 16) | (outs[i].validity ? 1u << 24 : 0u));
}

inline std::vector<uint32_t> signature(const ProjectArgs &A) {
  std::vector<uint32_t> s{SIG_PROJECT};
  sig_program(A.prog, s);
  sig_outs(A.outs, A.n_outs, s);
  return s;
}

inline std::vector<uint32_t> signature(const FilterArgs &A) {
  std::vector<uint32_t> s{SIG_FILTER};
  sig_program(A.prog, s);
  sig_outs(A.outs, A.n_outs, s);
  s.push_back(A.pred_pc);
  s.push_back(A.pred_cache_mask);
  s.push_back(A.out_cache_mask);
  return s;
}

inline std::vector<uint32_t> signature(const AggArgs &A) {
  std::vector<uint32_t> s{SIG_AGGREGATE};
  sig_program(A.prog, s);
  s.push_back((uint32_t)A.n_tasks);
  for (int i = 0; i < A.n_tasks; ++i)
    s.push_back(A.tasks[i].pc | ((uint32_t)A.tasks[i].role << 16) | ((uint32_t)A.tasks[i].idx << 20) | ((uint32_t)A.tasks[i].has_eval << 28));
  s.push_back((uint32_t)A.n_keys);
  for (int i = 0; i < A.n_keys; ++i) s.push_back(A.keys[i].type | ((uint32_t)A.keys[i].shift << 8) | ((uint32_t)A.keys[i].bits << 16));
  s.push_back((uint32_t)A.n_accs);
  for (int i = 0; i < A.n_accs; ++i)
    s.push_back(A.accs[i].op | ((uint32_t)A.accs[i].in_type << 8) | ((uint32_t)A.accs[i].track_nn << 16) | ((uint32_t)A.accs[i].has_input << 24));
  return s;
}

// Can a launch be served by a shape-specialised kernel at all?  Those kernels implement every
// instruction except the utf8 comparisons.
inline bool program_is_specialisable(const VmProgram &P) {
  for (int i = 0; i < VM_MAX_INSN; ++i)
    if (!vm_op_specialisable(P.insn[i] & 0xffu)) return false;
  return true;
}

}  // namespace qe
