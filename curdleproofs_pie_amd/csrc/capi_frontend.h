// C ABI, part 6: the shuffle verifier's front-end on the device (block program construction, launches).
// Part of the single translation unit csrc/msm_gpu.hip (included there, in this order; not a stand-alone header).
#pragma once

// ---------------------------------------------------------------- the shuffle verifier's front-end on the device (kernels_frontend.h)
struct cg1_shuffle_fe {
  int device = 0;
  cg1fe::Params pr{};
  uint32_t nops = 0, nlabels = 0;
  void *d_init = nullptr, *d_ops = nullptr, *d_labels = nullptr, *d_consts = nullptr, *d_tabG = nullptr, *d_tabH = nullptr;
  void *d_four = nullptr, *d_scratch = nullptr; size_t cap_n = 0;
  // the block program (kernels_frontend.h, second form): row descriptors per (node, word); the rows of a launch; passes per wave
  void *d_desc = nullptr, *d_rows = nullptr, *d_passes = nullptr; uint32_t n_nodes = 0; size_t cap_rows = 0, cap_blocks = 0, last_blocks = 0; uint32_t last_split[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  char err[200] = {0};
};

namespace {
// The verifier's transcript as an operation list: the mirror of prepare_one (csrc/shuffle_verify.cpp), which follows
// curdleproofs.py:176-180, same_perm.py:91-96, grand_prod.py:137-143, ipa.py:204-212 + :168-176, same_scalar.py:82-99,
// same_msm.py:194-206 + :158-173.  Challenges land in their slots of the row-input block (cg1rows::RowIn).
struct FeProgram {
  std::vector<cg1merlin::COp> ops;
  std::vector<uint32_t> table;
  std::vector<std::string> labels;
  uint32_t label(const char* s) {
    for (size_t i = 0; i < labels.size(); ++i) if (labels[i] == s) return (uint32_t)i;
    labels.emplace_back(s);
    uint8_t padded[32] = {0};
    memcpy(padded, s, strlen(s));
    for (int j = 0; j < 8; ++j) { uint32_t v; memcpy(&v, padded + 4 * j, 4); table.push_back(v); }
    return (uint32_t)labels.size() - 1;
  }
  void op(uint8_t kind, const char* lab, uint32_t len, uint32_t data_off, uint32_t out_off) {
    const uint32_t li = lab ? label(lab) : 0u, ll = lab ? (uint32_t)strlen(lab) : 0u;
    ops.push_back(cg1merlin::COp{(uint32_t)kind | (li << 8) | (ll << 16), len, data_off, out_off});
  }
  void point(const char* lab, size_t idx) { op(cg1merlin::OP_APPEND_POINT, lab, 48, (uint32_t)(idx * 48), 0); }
  void out(const char* lab, size_t slot, uint32_t len) { op(cg1merlin::OP_APPEND_OUT, lab, len, 0, (uint32_t)(slot * 32)); }
  void cst(const char* lab, uint32_t off) { op(cg1merlin::OP_APPEND_CONST, lab, 48, off, 0); }
  void challenge(const char* lab, size_t slot) { op(cg1merlin::OP_CHALLENGE_SCALAR, lab, 32, 0, (uint32_t)(slot * 32)); }
};

void fe_build_program(size_t ell, size_t lg, FeProgram& P) {
  const cg1rows::RowIn R{ell, lg};
  const size_t K = R.count(), base = 4 * ell;
  // own-point indices (csrc/shuffle_verify.cpp Layout)
  const size_t M = base, A = base + 1, T1 = base + 2, T2 = base + 3, U1 = base + 4, U2 = base + 5, Rp = base + 6, Sp = base + 7, B = base + 8, C = base + 9,
               Bc = base + 10, Bd = base + 11, LC = base + 12, RC = LC + lg, LD = LC + 2 * lg, RD = LC + 3 * lg, cmA1 = base + 12 + 4 * lg,
               Ba = cmA1 + 4, Bt = cmA1 + 5, Bu = cmA1 + 6, LA = cmA1 + 7, LT = LA + lg, LU = LA + 2 * lg, RA = LA + 3 * lg, RT = LA + 4 * lg, RU = LA + 5 * lg;
  for (size_t i = 0; i < 4 * ell; ++i) P.point("curdleproofs_step1", i);
  P.point("curdleproofs_step1", M);
  for (size_t i = 0; i < ell; ++i) P.challenge("curdleproofs_vec_a", R.a() + i);
  P.point("same_perm_step1", A); P.point("same_perm_step1", M);
  for (size_t i = 0; i < ell; ++i) P.out("same_perm_step1", R.a() + i, 32);
  P.challenge("same_perm_alpha", R.head() + 0); P.challenge("same_perm_beta", R.head() + 1);
  P.op(cg1fe::X_GPROD, nullptr, 0, 0, 0);
  P.point("gprod_step1", B); P.out("gprod_step1", K + 0, 32);
  P.challenge("gprod_alpha", R.head() + 2);
  P.point("gprod_step2", C); P.out("gprod_step2", K + 1, 32);
  P.challenge("gprod_beta", R.head() + 3);
  P.op(cg1fe::X_DA, nullptr, 0, 0, 0);
  P.point("ipa_step1", C); P.out("ipa_step1", K + 2, 48); P.out("ipa_step1", R.inner_prod(), 32); P.point("ipa_step1", Bc); P.point("ipa_step1", Bd);
  P.challenge("ipa_alpha", R.head() + 4); P.challenge("ipa_beta", R.head() + 5);
  for (size_t j = 0; j < lg; ++j) {
    P.point("ipa_loop", LC + j); P.point("ipa_loop", LD + j); P.point("ipa_loop", RC + j); P.point("ipa_loop", RD + j);
    P.challenge("ipa_gamma", R.gam() + j);
  }
  {
    const size_t order[10] = {Rp, Sp, T1, T2, U1, U2, cmA1, cmA1 + 1, cmA1 + 2, cmA1 + 3};
    for (size_t k = 0; k < 10; ++k) P.point("sameexp_points", order[k]);
  }
  P.challenge("same_scalar_alpha", R.head() + 6);
  P.out("same_msm_step1", K + 4, 48); P.point("same_msm_step1", T2); P.point("same_msm_step1", U2);
  for (size_t i = 0; i < ell; ++i) P.point("same_msm_step1", 2 * ell + i);
  P.cst("same_msm_step1", 0); P.cst("same_msm_step1", 0); P.cst("same_msm_step1", 48); P.cst("same_msm_step1", 0);       // Z Z H Z
  for (size_t i = 0; i < ell; ++i) P.point("same_msm_step1", 3 * ell + i);
  P.cst("same_msm_step1", 0); P.cst("same_msm_step1", 0); P.cst("same_msm_step1", 0); P.cst("same_msm_step1", 48);       // Z Z Z H
  P.point("same_msm_step1", Ba); P.point("same_msm_step1", Bt); P.point("same_msm_step1", Bu);
  P.challenge("same_msm_alpha", R.head() + 7);
  for (size_t j = 0; j < lg; ++j) {
    P.point("same_msm_loop", LA + j); P.point("same_msm_loop", LT + j); P.point("same_msm_loop", LU + j);
    P.point("same_msm_loop", RA + j); P.point("same_msm_loop", RT + j); P.point("same_msm_loop", RU + j);
    P.challenge("same_msm_gamma", R.gm() + j);
  }
  P.op(cg1fe::X_FINAL, nullptr, 0, 0, 0);
}

// The operation list cut into the nodes of kernels_frontend.h's block program: a symbolic run of STROBE (strobe.py:55-107) and of
// Merlin's framing (merlin_transcript.py:11-24, curdleproofs_transcript.py:15-25) that keeps, per byte of the sponge's rate, the
// constant XOR-ed into it and / or the place the byte comes from.  false = the program does not fit the row format (more than
// MAX_PIECES late pieces in a node, an offset too large): the caller keeps the byte-machine kernel.
struct FeNodes {
  struct Byte { uint8_t kind = 0; uint32_t src = 0; };              // 0 none, 1 byte of the lane's data row (src = its offset), 2 the challenge just drawn, 3 the out row
  struct Node { uint8_t T[168]; Byte D[168]; uint32_t type = cg1fe::N_PLAIN, bar = 0, da = 1, dr = 0, out_off = 0, len = 0; Node() { memset(T, 0, sizeof T); } };
  std::vector<Node> nodes;
  Node cur;
  uint32_t pos = 0, pos_begin = 0, cur_flags = 0;
  bool ok = true;
  int last_closed = -1;

  void run_f() {
    cur.T[pos] ^= (uint8_t)pos_begin; cur.T[pos + 1] ^= 0x04; cur.T[cg1merlin::STROBE_R + 1] ^= 0x80;
    nodes.push_back(cur);
    last_closed = (int)nodes.size() - 1;
    cur = Node();
    pos = 0; pos_begin = 0;
  }
  void put(uint8_t v) { cur.T[pos] ^= v; if (++pos == (uint32_t)cg1merlin::STROBE_R) run_f(); }
  void put_src(uint8_t kind, uint32_t src) { cur.D[pos].kind = kind; cur.D[pos].src = src; if (++pos == (uint32_t)cg1merlin::STROBE_R) run_f(); }
  void begin_op(uint8_t flags) {
    const uint32_t old = pos_begin;
    pos_begin = pos + 1;
    cur_flags = flags;
    put((uint8_t)old); put(flags);
    if ((flags & (cg1merlin::FLAG_C | cg1merlin::FLAG_K)) && pos != 0) run_f();
  }
  void frame(const std::string& label, uint32_t len) {
    begin_op(cg1merlin::FLAG_M | cg1merlin::FLAG_A);
    for (char c : label) put((uint8_t)c);
    for (int j = 0; j < 4; ++j) put((uint8_t)(len >> (8 * j)));
  }
  void barrier(uint32_t kind) { if (cur.bar) ok = false; cur.bar = kind; }
};

bool build_block_program(const std::vector<cg1merlin::COp>& ops, const std::vector<std::string>& labels, const uint8_t* init, const uint8_t* consts, bool generic,
                         std::vector<cg1merlin::RowDesc>& desc, uint32_t& n_nodes) {
  using namespace cg1merlin;
  FeNodes S;
  S.pos = init[200]; S.pos_begin = init[201]; S.cur_flags = init[202];
  if (S.pos >= (uint32_t)STROBE_R) return false;
  const uint32_t max_pieces = generic ? 4u : MAX_PIECES;            // (generic rows keep word 47 for the out-row offset / the final position)
  for (const COp& op : ops) {
    const uint32_t kind = op.kind_label & 0xffu, lab = (op.kind_label >> 8) & 0xffu, llen = op.kind_label >> 16;
    if (kind >= OP_BARRIER) {
      if (generic) return false;
      S.barrier(kind == cg1fe::X_GPROD ? 1u : (kind == cg1fe::X_DA ? 2u : 3u));
      continue;
    }
    if (lab >= labels.size()) return false;
    const std::string label = labels[lab].substr(0, llen);
    if (kind == OP_CHALLENGE_SCALAR || kind == OP_CHALLENGE) {
      const bool scalar = kind == OP_CHALLENGE_SCALAR;
      const uint32_t len = scalar ? 32u : op.len;
      if (len > 164u || (op.out_off & 3u) || (!scalar && !generic)) return false;
      S.frame(label, len);
      S.begin_op(FLAG_I | FLAG_A | FLAG_C);                              // the permutation the C flag forces closes the node the draw follows
      if (S.pos != 0 || S.last_closed < 0) return false;
      FeNodes::Node& sq = S.nodes[S.last_closed];
      if (sq.type != N_PLAIN) return false;
      sq.out_off = op.out_off; sq.len = len;
      if (!scalar) { sq.type = N_SQUEEZE_RAW; sq.da = 1; sq.dr = 0; S.pos = len; S.pos_begin = 0; continue; }
      sq.type = N_SQUEEZE; sq.da = 2; sq.dr = 1;
      // the redo node: the same frame and PRF header from (pos, pos_begin) = (32, 0), where every draw leaves the sponge
      S.pos = 32; S.pos_begin = 0;
      const size_t before = S.nodes.size();
      S.frame(label, 32);
      S.begin_op(FLAG_I | FLAG_A | FLAG_C);
      if (S.nodes.size() != before + 1 || S.pos != 0) return false;
      FeNodes::Node& rd = S.nodes.back();
      rd.type = N_SQUEEZE; rd.da = 1; rd.dr = 0; rd.out_off = op.out_off; rd.len = 32;
      // accepted: append_message(label, the 32 bytes), again from (32, 0)
      S.pos = 32; S.pos_begin = 0;
      S.frame(label, 32);
      S.begin_op(FLAG_A);
      for (uint32_t k = 0; k < 32; ++k) S.put_src(2, k);
      continue;
    }
    if (kind != OP_APPEND && kind != OP_APPEND_POINT && kind != OP_APPEND_CONST && kind != OP_APPEND_OUT) return false;
    if (kind == OP_APPEND_CONST && !consts) return false;
    S.frame(label, op.len);
    S.begin_op(FLAG_A);
    for (uint32_t k = 0; k < op.len; ++k) {
      if (kind == OP_APPEND_CONST) S.put(consts[op.data_off + k]);
      else if (kind == OP_APPEND_OUT) S.put_src(3, op.out_off + k);
      else S.put_src(1, op.data_off + k);
    }
  }
  S.cur.type = N_END;                                                     // what is left in the open node is never permuted
  S.cur.out_off = S.pos | (S.pos_begin << 8) | (S.cur_flags << 16);
  S.nodes.push_back(S.cur);
  if (!S.ok) return false;
  n_nodes = (uint32_t)S.nodes.size();
  desc.assign((size_t)n_nodes * ROW_WORDS, RowDesc{0, 0});
  for (uint32_t nd = 0; nd < n_nodes; ++nd) {
    const FeNodes::Node& N = S.nodes[nd];
    RowDesc* row = desc.data() + (size_t)nd * ROW_WORDS;
    for (uint32_t j = 0; j < 42; ++j) {
      uint32_t tw = 0;
      for (int b = 0; b < 4; ++b) tw |= (uint32_t)N.T[4 * j + b] << (8 * b);
      row[j].tword = tw;
      // data-row bytes of this word: one run of consecutive source bytes (a message is framed by >= 8 constant bytes)
      int lo = -1, cnt = 0;
      for (int b = 0; b < 4; ++b) if (N.D[4 * j + b].kind == 1) { if (lo < 0) lo = b; ++cnt; }
      if (cnt) {
        const uint32_t s0 = N.D[4 * j + lo].src;
        for (int b = 0; b < cnt; ++b) if (N.D[4 * j + lo + b].kind != 1 || N.D[4 * j + lo + b].src != s0 + b) return false;
        if (s0 >= (1u << 27) || (!generic && s0 % 48u + cnt > 48u)) return false;
        row[j].src = 1u | ((uint32_t)lo << 1) | ((uint32_t)(cnt - 1) << 3) | (s0 << 5);
      }
    }
    if (generic) {
      row[42].tword = N.type | (N.da << 4) | (N.dr << 6) | ((N.type == N_SQUEEZE_RAW ? N.len : 0u) << 8);
      row[47].tword = N.out_off;
    } else {
      if (N.type != N_END && ((N.out_off & 31u) || (N.out_off >> 5) >= (1u << 16))) return false;
      row[42].tword = N.type == N_END ? (N_END | (N.bar << 2)) : (N.type | (N.bar << 2) | (N.da << 4) | (N.dr << 6) | ((N.out_off >> 5) << 8));
    }
    if (N.type == N_END && !generic) continue;
    // late pieces: runs of bytes of kind 2 / 3 with consecutive sources
    uint32_t np = 0;
    for (uint32_t p = 0; p < (uint32_t)STROBE_R;) {
      const uint8_t k = N.D[p].kind;
      if (k < 2) { ++p; continue; }
      uint32_t len = 1;
      while (p + len < (uint32_t)STROBE_R && len < 48u && N.D[p + len].kind == k && N.D[p + len].src == N.D[p].src + len) ++len;
      if (np == max_pieces || N.D[p].src >= (1u << 17)) return false;
      row[43 + np].tword = len | (p << 6) | ((k == 3 ? 1u : 0u) << 14) | (N.D[p].src << 15);
      ++np;
      p += len;
    }
  }
  return true;
}
bool fe_build_nodes(const FeProgram& P, const uint8_t* init, const uint8_t* consts, std::vector<cg1fe::RowDesc>& desc, uint32_t& n_nodes) {
  return build_block_program(P.ops, P.labels, init, consts, false, desc, n_nodes);
}
}  // namespace

extern "C" {

void cg1_shuffle_fe_destroy(cg1_shuffle_fe* fe) {
  if (!fe) return;
  (void)hipSetDevice(fe->device);
  for (void* p : {fe->d_init, fe->d_ops, fe->d_labels, fe->d_consts, fe->d_tabG, fe->d_tabH, fe->d_four, fe->d_scratch, fe->d_desc, fe->d_rows, fe->d_passes})
    if (p) (void)hipFree(p);
  delete fe;
}

// crs_affine96 / crs48: the ell + 9 CRS points (crs.py:92-101 order) decoded and as they stand on the wire
cg1_shuffle_fe* cg1_shuffle_fe_create(cg1_ctx* ctx, size_t ell, size_t lg, const uint8_t* crs_affine96, const uint8_t* crs48) {
  if (!ctx || !crs_affine96 || !crs48 || ell == 0 || lg == 0 || lg > 20 || ((ell + 4) != ((size_t)1 << lg))) return nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
  cg1_shuffle_fe* fe = new cg1_shuffle_fe();
  fe->device = ctx->device;
  const cg1rows::RowIn R{ell, lg};
  cg1fe::Params& pr = fe->pr;
  pr.ell = (uint32_t)ell; pr.lg = (uint32_t)lg; pr.L = (uint32_t)(4 * ell + 19 + 10 * lg); pr.K = (uint32_t)R.count();
  pr.out_stride = (pr.K + 6u) * 32u;
  pr.idx_A = (uint32_t)(4 * ell + 1); pr.idx_T1 = (uint32_t)(4 * ell + 2); pr.idx_U1 = (uint32_t)(4 * ell + 4); pr.idx_B = (uint32_t)(4 * ell + 8);
  pr.idx_T0 = (uint32_t)(2 * ell);
  FeProgram P;
  fe_build_program(ell, lg, P);
  if (P.labels.size() > (size_t)cg1merlin::MAX_LABELS) { delete fe; return nullptr; }
  fe->nops = (uint32_t)P.ops.size(); fe->nlabels = (uint32_t)P.labels.size();
  uint8_t init[CG1_MERLIN_STATE_BYTES];
  cg1_merlin_init(init, (const uint8_t*)"curdleproofs", 12);                      // CurdleproofsTranscript(b"curdleproofs"), curdleproofs.py:172
  uint8_t consts[96];
  memset(consts, 0, sizeof consts);
  consts[0] = 0xC0;                                                               // Z1 as the wheel serialises it
  memcpy(consts + 48, crs48 + (ell + 4) * 48, 48);                                // crs.H
  bool ok = hipMalloc(&fe->d_init, sizeof init) == hipSuccess && hipMalloc(&fe->d_ops, P.ops.size() * sizeof(cg1merlin::COp)) == hipSuccess &&
            hipMalloc(&fe->d_labels, P.table.size() * 4) == hipSuccess && hipMalloc(&fe->d_consts, sizeof consts) == hipSuccess &&
            hipMalloc(&fe->d_tabG, 8192 * sizeof(cg1::PreparedPoint)) == hipSuccess && hipMalloc(&fe->d_tabH, 8192 * sizeof(cg1::PreparedPoint)) == hipSuccess;
  ok = ok && hipMemcpy(fe->d_init, init, sizeof init, hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(fe->d_ops, P.ops.data(), P.ops.size() * sizeof(cg1merlin::COp), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(fe->d_labels, P.table.data(), P.table.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(fe->d_consts, consts, sizeof consts, hipMemcpyHostToDevice) == hipSuccess;
  if (ok) {
    std::vector<cg1fe::RowDesc> desc;
    uint32_t nn = 0;
    if (fe_build_nodes(P, init, consts, desc, nn)) {
      ok = hipMalloc(&fe->d_desc, desc.size() * sizeof(cg1fe::RowDesc)) == hipSuccess &&
           hipMemcpy(fe->d_desc, desc.data(), desc.size() * sizeof(cg1fe::RowDesc), hipMemcpyHostToDevice) == hipSuccess;
      fe->n_nodes = nn;
    }
  }
  // fixed-base tables: entry [w][b] = b * 2^(8 w) * base for the two bases of D (grand_prod.py:157), as 128-byte Montgomery records
  if (ok) {
    std::vector<uint8_t> sc(8192 * 32, 0);
    for (int w = 0; w < 32; ++w) for (int b = 0; b < 256; ++b) sc[((size_t)w * 256 + b) * 32 + w] = (uint8_t)b;
    DevBuf dsc, dbase, dout, dflags;
    ok = dsc.alloc(sc.size()) == hipSuccess && dbase.alloc(96) == hipSuccess && dout.alloc(8192 * 96) == hipSuccess && dflags.alloc(8192 + 16) == hipSuccess &&
         hipMemcpy(dsc.p, sc.data(), sc.size(), hipMemcpyHostToDevice) == hipSuccess;
    for (int which = 0; which < 2 && ok; ++which) {
      const uint8_t* src = crs_affine96 + (ell + 4 + 3 + which) * 96;              // G_sum, H_sum
      ok = hipMemcpy(dbase.p, src, 96, hipMemcpyHostToDevice) == hipSuccess &&
           cg1_batch_mul_device(ctx, dbase.p, 1, dsc.p, dout.p, 8192) == CG1_OK;
      if (ok) {
        hipLaunchKernelGGL(cg1::k_prepare_points, dim3(32), dim3(256), 0, ctx->stream, (const uint32_t*)dout.p,
                           (cg1::PreparedPoint*)(which ? fe->d_tabH : fe->d_tabG), (uint8_t*)dflags.p, 8192u);
        ok = hipStreamSynchronize(ctx->stream) == hipSuccess && hipGetLastError() == hipSuccess;
      }
    }
  }
  if (!ok) { cg1_shuffle_fe_destroy(fe); return nullptr; }
  return fe;
}

size_t cg1_shuffle_fe_aux_bytes(void) { return 19 * 32; }

// Host only (no GPU needed; test support): ONE transcript of cg1_merlin_batch_device's interface run through the block program on the
// CPU -- build_block_program's tables walked as k_fill_rows + k_merlin_batch_rows walk them (rows, late pieces, first-draw / redo /
// raw squeeze nodes, the open block at the end).  Returns CG1_ERR_ARG when the operation list does not fit the row format (the device
// entry point then serves the call with the byte-level machine).  state_out208 may be NULL.
int cg1_merlin_block_program_emulate(const uint8_t* init_state208, const cg1_merlin_op* ops, size_t nops, const uint8_t* data_row, size_t data_bytes,
                                     uint8_t* out_row, size_t out_bytes, uint8_t* state_out208, uint32_t* passes) {
  if (!init_state208 || (nops && !ops) || !out_row) return CG1_ERR_ARG;
  std::vector<cg1merlin::COp> cops(nops);
  std::vector<std::string> labels;
  for (size_t k = 0; k < nops; ++k) {
    const cg1_merlin_op& o = ops[k];
    if (o.kind > 3 || o.label_len > 32) return CG1_ERR_ARG;
    if (o.kind == 0 && (size_t)o.data_off + o.len > data_bytes) return CG1_ERR_ARG;
    if (o.kind != 0 && (size_t)o.out_off + (o.kind == 2 ? 32 : o.len) > out_bytes) return CG1_ERR_ARG;
    const std::string lb((const char*)o.label, o.label_len);
    size_t idx = 0;
    while (idx < labels.size() && labels[idx] != lb) ++idx;
    if (idx == labels.size()) { if (labels.size() >= (size_t)cg1merlin::MAX_LABELS) return CG1_ERR_ARG; labels.push_back(lb); }
    cops[k] = cg1merlin::COp{(uint32_t)o.kind | ((uint32_t)idx << 8) | ((uint32_t)o.label_len << 16), o.len, o.data_off, o.out_off};
  }
  std::vector<cg1merlin::RowDesc> desc;
  uint32_t nn = 0;
  if (!build_block_program(cops, labels, init_state208, nullptr, true, desc, nn)) return CG1_ERR_ARG;
  uint8_t sponge[200], drawn[36] = {0};
  memcpy(sponge, init_state208, 200);
  uint32_t nd = 0, np = 0;
  for (;;) {
    const cg1merlin::RowDesc* row = desc.data() + (size_t)nd * cg1merlin::ROW_WORDS;
    const uint32_t info = row[42].tword, type = info & 3u, aux = row[47].tword;
    for (uint32_t q = 0; q < 4; ++q) {
      const uint32_t pc = row[43 + q].tword, len = pc & 63u;
      if (!len) continue;
      const uint32_t dst = (pc >> 6) & 255u, from_row = (pc >> 14) & 1u, so = pc >> 15;
      for (uint32_t i = 0; i < len; ++i) sponge[dst + i] ^= from_row ? out_row[so + i] : drawn[so + i];
    }
    for (uint32_t j = 0; j < 42; ++j) {
      uint32_t v = row[j].tword;
      if (row[j].src) {
        const uint32_t lo = (row[j].src >> 1) & 3u, cnt = ((row[j].src >> 3) & 3u) + 1u, off = row[j].src >> 5;
        for (uint32_t b = 0; b < cnt; ++b) v ^= (uint32_t)data_row[off + b] << (8 * (lo + b));
      }
      for (int b = 0; b < 4; ++b) sponge[4 * j + b] ^= (uint8_t)(v >> (8 * b));
    }
    if (type == cg1merlin::N_END) {
      if (state_out208) { memcpy(state_out208, sponge, 200); state_out208[200] = (uint8_t)aux; state_out208[201] = (uint8_t)(aux >> 8); state_out208[202] = (uint8_t)(aux >> 16); memset(state_out208 + 203, 0, 5); }
      break;
    }
    cg1_keccak_f1600(sponge);
    ++np;
    bool accept = true;
    if (type == cg1merlin::N_SQUEEZE) {
      uint8_t dv[32];
      memcpy(dv, sponge, 32);
      memset(sponge, 0, 32);
      cg1fr::fr tmp;
      bool nonzero = false;
      for (int i = 0; i < 32; ++i) nonzero |= dv[i] != 0;
      accept = nonzero && cg1fr::fr_from_le32(dv, tmp);
      if (accept) { memcpy(out_row + aux, dv, 32); memcpy(drawn, dv, 32); }
    } else if (type == cg1merlin::N_SQUEEZE_RAW) {
      const uint32_t len = (info >> 8) & 0xffu;
      memcpy(out_row + aux, sponge, len);
      memset(sponge, 0, len);
    }
    nd += accept ? (info >> 4) & 3u : (info >> 6) & 3u;
  }
  if (passes) *passes = np;
  return CG1_OK;
}

// Host only (no GPU needed; test support): walk the block program of one proof on the CPU exactly as k_shuffle_front_end_rows does --
// rows as k_fill_rows builds them, late pieces, both kinds of squeeze node -- up to the first barrier step (the grand product),
// and return the out row (the challenges drawn so far sit in their slots: vec_a, alpha, beta of same_perm).  wire = the proof's L own
// points as cg1_shuffle_gather_points packs them; out_row: (K + 6) * 32 bytes, zero where nothing was drawn.  *passes = permutations.
int cg1_shuffle_fe_emulate_to_first_barrier(size_t ell, size_t lg, const uint8_t* crs_h48, const uint8_t* wire, uint8_t* out_row, size_t out_row_bytes,
                                            uint32_t* passes) {
  if (!wire || !out_row || !crs_h48 || ell == 0 || lg == 0 || lg > 20 || ((ell + 4) != ((size_t)1 << lg))) return CG1_ERR_ARG;
  const cg1rows::RowIn R{ell, lg};
  if (out_row_bytes < (R.count() + 6) * 32) return CG1_ERR_ARG;
  FeProgram P;
  fe_build_program(ell, lg, P);
  uint8_t init[CG1_MERLIN_STATE_BYTES], consts[96];
  cg1_merlin_init(init, (const uint8_t*)"curdleproofs", 12);
  memset(consts, 0, sizeof consts);
  consts[0] = 0xC0;
  memcpy(consts + 48, crs_h48, 48);
  std::vector<cg1fe::RowDesc> desc;
  uint32_t nn = 0;
  if (P.labels.size() > (size_t)cg1merlin::MAX_LABELS || !fe_build_nodes(P, init, consts, desc, nn)) return CG1_ERR_ARG;
  memset(out_row, 0, out_row_bytes);
  uint8_t sponge[200], drawn[32] = {0};
  memcpy(sponge, init, 200);
  uint32_t nd = 0, np = 0;
  for (;;) {
    const cg1fe::RowDesc* row = desc.data() + (size_t)nd * cg1fe::ROW_WORDS;
    const uint32_t info = row[42].tword, type = info & 3u;
    if (((info >> 2) & 3u) != 0u || type == cg1fe::N_END) break;
    for (uint32_t q = 0; q < cg1fe::MAX_PIECES; ++q) {
      const uint32_t pc = row[43 + q].tword, len = pc & 63u;
      if (!len) break;
      const uint32_t dst = (pc >> 6) & 255u, from_row = (pc >> 14) & 1u, so = pc >> 15;
      for (uint32_t i = 0; i < len; ++i) sponge[dst + i] ^= from_row ? out_row[so + i] : drawn[so + i];
    }
    for (uint32_t j = 0; j < 42; ++j) {                      // the row as k_fill_rows writes it
      uint32_t v = row[j].tword;
      if (row[j].src) {
        const uint32_t lo = (row[j].src >> 1) & 3u, cnt = ((row[j].src >> 3) & 3u) + 1u, off = row[j].src >> 5, k0 = off % 48u;
        const uint8_t* pt = wire + (size_t)(off - k0);
        const bool inf = (pt[0] & 0xC0u) == 0xC0u;
        for (uint32_t b = 0; b < cnt; ++b) v ^= (uint32_t)(inf ? (k0 + b == 0 ? 0xC0u : 0u) : pt[k0 + b]) << (8 * (lo + b));
      }
      for (int b = 0; b < 4; ++b) sponge[4 * j + b] ^= (uint8_t)(v >> (8 * b));
    }
    cg1_keccak_f1600(sponge);
    ++np;
    bool accept = true;
    if (type == cg1fe::N_SQUEEZE) {
      uint8_t dv[32];
      memcpy(dv, sponge, 32);
      memset(sponge, 0, 32);
      cg1fr::fr tmp;
      bool nonzero = false;
      for (int i = 0; i < 32; ++i) nonzero |= dv[i] != 0;
      accept = nonzero && cg1fr::fr_from_le32(dv, tmp);
      if (accept) { memcpy(out_row + 32 * ((info >> 8) & 0xffffu), dv, 32); memcpy(drawn, dv, 32); }
    }
    nd += accept ? (info >> 4) & 3u : (info >> 6) & 3u;
  }
  if (passes) *passes = np;
  return CG1_OK;
}

// Host only (no GPU needed): the shape of the block program for a given ell -- operations of the verifier's transcript, nodes they are
// cut into (0: does not fit the row format), squeeze nodes (two per challenge: first draw and redo) and the largest number of late
// pieces in a node.  out4 = {operations, nodes, squeeze nodes, max pieces}.
int cg1_shuffle_fe_program_shape(size_t ell, size_t lg, uint32_t* out4) {
  if (!out4 || ell == 0 || lg == 0 || lg > 20 || ((ell + 4) != ((size_t)1 << lg))) return CG1_ERR_ARG;
  FeProgram P;
  fe_build_program(ell, lg, P);
  uint8_t init[CG1_MERLIN_STATE_BYTES], consts[96];
  cg1_merlin_init(init, (const uint8_t*)"curdleproofs", 12);
  memset(consts, 0, sizeof consts);
  consts[0] = 0xC0;
  std::vector<cg1fe::RowDesc> desc;
  uint32_t nn = 0;
  out4[0] = (uint32_t)P.ops.size(); out4[1] = out4[2] = out4[3] = 0;
  if (P.labels.size() > (size_t)cg1merlin::MAX_LABELS || !fe_build_nodes(P, init, consts, desc, nn)) return CG1_OK;
  out4[1] = nn;
  for (uint32_t nd = 0; nd < nn; ++nd) {
    const cg1fe::RowDesc* row = desc.data() + (size_t)nd * cg1fe::ROW_WORDS;
    if ((row[42].tword & 3u) == cg1fe::N_SQUEEZE) ++out4[2];
    uint32_t np = 0;
    for (uint32_t q = 0; q < cg1fe::MAX_PIECES; ++q) if (row[43 + q].tword & 63u) ++np;
    out4[3] = std::max(out4[3], np);
  }
  return CG1_OK;
}

// Nodes of the block program (0: the program of this ell does not fit the row format and the byte-machine kernel is used).
size_t cg1_shuffle_fe_nodes(const cg1_shuffle_fe* fe) { return fe ? fe->n_nodes : 0; }

// Keccak passes of the slowest wave of the last launch enqueued on `ctx` (waits for the stream; 0 if that launch used the byte machine).
size_t cg1_shuffle_fe_last_passes(cg1_shuffle_fe* fe, cg1_ctx* ctx) {
  if (!fe || !ctx || !fe->last_blocks || !fe->d_passes) return 0;
  if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return 0;
  std::vector<uint32_t> h(fe->last_blocks * 8);
  if (hipMemcpy(h.data(), fe->d_passes, h.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  size_t best = 0;
  for (size_t b = 0; b < fe->last_blocks; ++b) if (h[8 * b] > h[8 * best]) best = b;
  for (int k = 0; k < 8; ++k) fe->last_split[k] = h[8 * best + k];
  return h[8 * best];
}
// ... and that wave's shader-clock split (launches with cg1_ctx_set_param("fe_timed", 1)): out[0..7) = clocks / 256 spent in (late pieces +
// issuing the row loads, Keccak-f, whole passes, draw + range check, X_GPROD, X_DA, X_FINAL); valid after cg1_shuffle_fe_last_passes.
void cg1_shuffle_fe_last_split(const cg1_shuffle_fe* fe, uint32_t* out7) { for (int k = 0; k < 7; ++k) out7[k] = fe ? fe->last_split[k + 1] : 0; }

// Enqueue the front-end of n proofs on ctx's compute stream (no wait: cg1_stream_sync).  d_wire48: n x L own points as gathered from
// the wire (cg1_shuffle_gather_points); d_pts_affine96: the same points decoded (cg1_batch_decompress_*); d_aux: n x 19 x 32 bytes
// (cg1_shuffle_gather_aux); outputs as cg1_shuffle_prepare_inputs: d_rowin n x cg1_shuffle_rowin_scalars() x 32, d_status n codes.
int cg1_shuffle_fe_enqueue(cg1_shuffle_fe* fe, cg1_ctx* ctx, size_t n, const void* d_wire48, const void* d_pts_affine96, const void* d_aux,
                           void* d_rowin, void* d_status, int lanes_per_wave) {
  if (!fe || !ctx) return CG1_ERR_ARG;
  if (n == 0) return CG1_OK;
  if (!d_wire48 || !d_pts_affine96 || !d_aux || !d_rowin || !d_status || n >= (1u << 24) || ctx->device != fe->device) return CG1_ERR_ARG;
  if (lanes_per_wave < 1 || lanes_per_wave > cg1merlin::LANES) lanes_per_wave = cg1merlin::LANES;
  HIPCHK(hipSetDevice(ctx->device));
  if (n > fe->cap_n) {
    if (fe->d_four) (void)hipFree(fe->d_four);
    if (fe->d_scratch) (void)hipFree(fe->d_scratch);
    fe->d_four = fe->d_scratch = nullptr; fe->cap_n = 0;
    HIPCHK(hipMalloc(&fe->d_four, n * 4 * sizeof(cg1::PreparedPoint)));
    HIPCHK(hipMalloc(&fe->d_scratch, n * (size_t)fe->pr.out_stride));
    fe->cap_n = n;
  }
  hipLaunchKernelGGL(cg1fe::k_fe_gather4, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)d_pts_affine96, fe->pr, (uint32_t)n,
                     (cg1::PreparedPoint*)fe->d_four);
  const unsigned nblk = (unsigned)((n + lanes_per_wave - 1) / lanes_per_wave);
  fe->pr.prio = (uint32_t)ctx->fe_prio;
  if (fe->n_nodes && ctx->fe_rows) {
    const size_t row_words = (size_t)fe->n_nodes * cg1fe::ROW_WORDS;
    const size_t need = (size_t)nblk * lanes_per_wave * row_words;          // rows are laid out per wave: the last wave's unused lanes count
    if (need > fe->cap_rows || nblk > fe->cap_blocks) {
      if (fe->d_rows) (void)hipFree(fe->d_rows);
      if (fe->d_passes) (void)hipFree(fe->d_passes);
      fe->d_rows = fe->d_passes = nullptr; fe->cap_rows = 0; fe->cap_blocks = 0; fe->last_blocks = 0;
      HIPCHK(hipMalloc(&fe->d_rows, need * 4));
      HIPCHK(hipMalloc(&fe->d_passes, (size_t)nblk * 32));
      fe->cap_rows = need; fe->cap_blocks = nblk;
    }
    fe->last_blocks = nblk;
    hipLaunchKernelGGL(cg1merlin::k_fill_rows, dim3((unsigned)((row_words + 255) / 256), (unsigned)std::min<size_t>(n, 65535)), dim3(256), 0, ctx->stream, (const cg1fe::RowDesc*)fe->d_desc, fe->n_nodes,
                       (const uint8_t*)d_wire48, (size_t)fe->pr.L * 48, 1u, (uint32_t)n, (uint32_t)lanes_per_wave, (uint32_t*)fe->d_rows);
    hipLaunchKernelGGL(ctx->fe_timed ? cg1fe::k_shuffle_front_end_rows<true> : cg1fe::k_shuffle_front_end_rows<false>, dim3(nblk), dim3(cg1merlin::LANES), 0, ctx->stream, (const uint8_t*)fe->d_init, (const uint32_t*)fe->d_rows,
                       fe->n_nodes, (const uint8_t*)d_wire48, (const uint8_t*)d_aux, (const cg1::PreparedPoint*)fe->d_four, (const cg1::PreparedPoint*)fe->d_tabG,
                       (const cg1::PreparedPoint*)fe->d_tabH, fe->pr, (uint8_t*)fe->d_scratch, (uint8_t*)d_rowin, (int32_t*)d_status, (uint32_t)n,
                       (uint32_t)lanes_per_wave, (uint32_t*)fe->d_passes);
    HIPCHK(hipGetLastError());
    return CG1_OK;
  }
  fe->last_blocks = 0;
  hipLaunchKernelGGL(cg1fe::k_shuffle_front_end, dim3(nblk), dim3(cg1merlin::LANES), 0, ctx->stream, (const uint8_t*)fe->d_init, (const cg1merlin::COp*)fe->d_ops,
                     fe->nops, (const uint32_t*)fe->d_labels, fe->nlabels, (const uint8_t*)fe->d_consts, (const uint8_t*)d_wire48, (const uint8_t*)d_aux,
                     (const cg1::PreparedPoint*)fe->d_four, (const cg1::PreparedPoint*)fe->d_tabG, (const cg1::PreparedPoint*)fe->d_tabH, fe->pr,
                     (uint8_t*)fe->d_scratch, (uint8_t*)d_rowin, (int32_t*)d_status, (uint32_t)n, (uint32_t)lanes_per_wave);
  HIPCHK(hipGetLastError());
  return CG1_OK;
}

}  // extern "C"
