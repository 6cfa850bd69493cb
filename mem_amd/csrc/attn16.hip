// Attention for the 14 x 14 token grid (197 tokens: ViT-B/16 and ViT-L/16 at 224 x 224, the BASELINE configuration):
// forward, and a FUSED backward that computes the probabilities once.  Same contract, rounding points and outputs as the
// general kernels in attn.hip (reference: Attention.forward, mem/modeling_finetune.py:137-154, bias of RelativePositionBias
// :213-247); attn.hip dispatches here when the window is 14 x 14.
//
// What the general kernels spend their time on (profiles/r02: 9-15 % MFMA-busy, VALU-issue bound): per score element one
// integer subtract + one LDS gather for the bias bucket, 1.5 instructions to unpack the bf16-rounded score, and -- in
// backward -- the whole softmax recomputation TWICE (once per operand orientation: dK/dV need dS with queries down the
// accumulator rows, dQ needs keys down the rows).  Here:
//
//   * KEY SLOTS.  The K / V images in LDS hold the keys in "slot" order: grid row ky occupies slots 16 ky .. 16 ky + 13,
//     slots 16 ky + 14, 15 are padding (zero rows), the cls token sits in slot 14 (the first padding slot of row 0):
//     224 slots = 7 blocks of 32.  A lane of the S^T = K Q^T accumulator (32x32 MFMA: lane = query column, registers =
//     key rows 8 g + 4 hh + e of the block) then sees, for register (kb, g, e), the key (ky, kx) = (2 kb + (g >> 1),
//     8 (g & 1) + 4 hh + e): the bias bucket (qy - ky + 13) * 27 + (qx - kx + 13) is a PER-LANE base minus a
//     COMPILE-TIME offset.  The head's table is stored reversed in LDS, so the bias of a register is
//     `ds_read_b32 base_lane offset:const` -- no index arithmetic, no code tables, reads are issued far ahead.
//     The cls query reads a constant region behind the table, the one cls key element and the padding slots are
//     selected on hh (wave-uniform positions).
//   * v_dot2c_f32_bf16 with the selector pair (1, 0) / (0, 1) adds the bias to the low / high half of a packed bf16 pair:
//     unpack + add in ONE instruction (S is rounded to bf16 as the reference's autocast q k^T output is, the sum is fp32).
//   * FUSED BACKWARD.  Wave w owns query block w AND key block w.  In step s it computes P^T, dS^T of the tile
//     (queries w, keys (w + s) mod 7) from its own Q / dO fragments, accumulates dQ^T straight out of the accumulators,
//     and publishes the two tiles as bf16 (2 x 2 KiB) in LDS; after a barrier it picks up the tile (queries (w - s) mod 7,
//     keys w) with transposing reads (ds_read_b64_tr_b16: the K dimension of the MFMA becomes the queries) and
//     accumulates dK^T, dV^T.  Every tile's softmax work is done once; 20 MFMAs per tile instead of 28; Q, K, V, dO are
//     read from HBM once instead of twice.  Four head images (112 KiB) + the exchange tiles (28 KiB) + tables fill the
//     160 KiB of LDS, so the images are single-buffered: the LDS-DMA of the next sample is issued as soon as the last
//     step has passed its barrier and runs under the sample's epilogue (stores, bucket fold).
//   * the bias-table gradient uses the same constant offsets for its fixed-point LDS atomics (ds_add_u32); the bound of
//     the fixed-point scale is computed per sample inside the kernel (no per-head statistics pass).  (query + 1, key + 1) is
//     the bucket of (query, key), so the four keys of a register group are summed along the diagonal over neighbouring
//     lanes of the whole wave (wave_shr:1) and a group costs 2 wave-wide atomics (round 4; an LDS atomic costs its wave
//     ~52 cycles whatever its lane count).
//     (Round 5: an EIGHTH wave that owns the table gradient -- reads the published dS tiles, does chains and adds -- was
//     measured and dropped: 374 us per layer against 268 -- one wave does not get through the 28 dependent DPP chains of a step
//     in the time the seven take for the step; with fp32 buckets (ds_add_f32, single writer) 670 us: a float LDS atomic takes
//     195 cycles, an integer one 11 (tools/micro/lds_atomic_lat.hip).  tools/exp/attn16_bucket_wave.patch, tools/exp/README.md.)
//   * FORWARD (round 4): eight waves -- the eighth only issues the LDS-DMA of the next sample (issue time, not memory
//     time, was 23 % of the seven-wave kernel), and waves 4-6 run half a sample behind waves 0-3 so that the two waves of a
//     SIMD alternate between their MFMA and VALU phases; K and V are double-buffered separately.
#include "attn_common.hpp"
#include <type_traits>

#define ATTN16_FD_POS 0   // where the fused-delta loads of the next sample are issued: 0 = in front of the Q / dO DMA (B = 256:
                          // 300.5 us per layer), 1 = behind dQ's LDS staging (306.5 us); attn_delta + unfused backward: 314.4 us
#define ATTN16_SKEW 1     // forward: waves 4..6 half a sample behind waves 0..3 (0: lockstep, one barrier per sample)
#define ATTN16_W3STAGE 0  // backward: 1 = wave 3 (alone on its SIMD) stages the Q / dO images of the next sample for everybody: measured 278 vs 266 us

namespace {

constexpr int W16 = 14;                                   // window height = width
constexpr int R16 = 2 * W16 - 1;                          // 27 buckets per relative row
constexpr int OFF16 = (W16 - 1) * R16 + (W16 - 1);        // 364
constexpr int M16 = 2 * OFF16;                            // 728: reversed index a = M16 - bucket
constexpr int KMAX16 = (W16 - 1) * R16 + 15;              // 366: largest key code, padding slots included
constexpr int CLSQ16 = M16 + 3;                           // 731: first entry of the constant region read by the cls query
constexpr int TABLEN16 = (CLSQ16 + KMAX16 + 1 + 3) & ~3;  // 1100 entries
constexpr int T16 = W16 * W16 + 1;                        // 197 tokens
constexpr int TP16 = 16 * W16;                            // 224 key slots / padded queries
constexpr int NB16 = TP16 / 32;                           // 7 blocks = 7 waves
constexpr int IMG16 = TP16 * 128;                         // bytes of a head image
constexpr int kThreads16 = NB16 * 64;
constexpr int NRD16 = R16 * R16 + 3;                      // 732 table rows

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
// c + (low / high bf16 of pk): v_dot2c_f32_bf16 with the selector pair (1, 0) / (0, 1).  Measured on gfx950
// (tools/micro/dot2_bf16.hip): the half selection is exact; the fp32 sum is TRUNCATED, not rounded to nearest (1 ulp of
// fp32 low in ~6 % of the cases) -- far below the bf16 rounding of the operands this kernel family works with (the fp32
// parity mode does not use these kernels).  The low selector must live in a REGISTER: written as a constant, hipcc
// (ROCm 7.2) encodes 0x00003F80 as the inline constant 1.0, which this instruction reads as 0x3F800000 = the HIGH half.
__device__ __forceinline__ unsigned sel_lo_reg() {
  unsigned v;
  asm volatile("s_mov_b32 %0, 0x3f80" : "=s"(v));
  return v;
}
__device__ __forceinline__ float add_lo(unsigned pk, float c, unsigned sel_lo) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pk), __builtin_bit_cast(bf16x2_t, sel_lo), c, false);
}
__device__ __forceinline__ float add_hi(unsigned pk, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pk), __builtin_bit_cast(bf16x2_t, 0x3F800000u), c, false);
}

// ---- coalesced row stores.  An accumulator tile holds a token per LANE and its 64 head-dim values down the registers: a
// direct store writes 8 bytes per lane, 4.5 KiB apart -- 64 partial cache-line writes per wave-instruction (measured:
// the stores of one sample took ~13 K cycles, a fifth of the fused backward).  The tile goes through a 4 KiB LDS slot of the
// wave instead ([32 tokens][128 B], 16-byte chunk c of token r at c ^ ((r >> 1) & 7)) and leaves as 16 bytes per lane,
// 8 lanes per 128-byte row.  Lanes whose row is padding write to a trash page, so that every wave issues every store
// (the counted vmcnt waits rely on the number of stores in flight).
__device__ __attribute__((aligned(256))) unsigned char g_attn16_trash[1024];
__device__ __forceinline__ void tile_put(char* st, int r, int hh, int db, int g, bf16x4 w) {
  const int c = db * 4 + g;
  *reinterpret_cast<bf16x4*>(st + r * 128 + ((c ^ ((r >> 1) & 7)) << 4) + hh * 8) = w;
}
__device__ __forceinline__ bf16x8 tile_get(const char* st, int row, int c) {
  return *reinterpret_cast<const bf16x8*>(st + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
}

// s_waitcnt vmcnt(n) through the builtin (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14): the
// compiler's waitcnt pass sees it and does not add a vmcnt(0) of its own in front of the first use of a plainly loaded
// register (which would wait for the stores as well)
#define ATTN16_WAIT_VM(n)                                                                \
  do {                                                                                   \
    __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14));  \
    asm volatile("" ::: "memory");                                                       \
  } while (0)

// token of a key slot (-1: padding)
__device__ __forceinline__ int slot_tok(int s) {
  const int kx = s & 15;
  return kx < W16 ? 1 + (s >> 4) * W16 + kx : (s == W16 ? 0 : -1);
}
// Stage a head slice in SLOT order (LDS image layout of attn_common.hpp, the "token" of the swizzle is the slot).
// (instructions first, first + step, ...: all waves share an image, or one wave stages it alone)
__device__ __forceinline__ void stage_slots(char* dst, const __bf16* src, long long ld, int first, int step) {
  const int lane = threadIdx.x & 63;
  for (int inst = first; inst < TP16 / 8; inst += step) {
    const int slot = inst * 8 + (lane >> 3), cpos = lane & 7;
    const int chunk = cpos ^ img_key(slot);
    const int tok = slot_tok(slot);
    const void* g = tok >= 0 ? (const void*)(src + (long long)tok * ld + chunk * 8)
                             : (const void*)(g_attn_zero_page + cpos * 16);
    glds16(g, dst + inst * 1024);
  }
}
__device__ __forceinline__ void stage_slots(char* dst, const __bf16* src, long long ld) {
  stage_slots(dst, src, ld, threadIdx.x >> 6, NB16);
}
// natural token order, 224 rows (tokens >= 197: zero rows)
__device__ __forceinline__ void stage_tokens(char* dst, const __bf16* src, long long ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int inst = wave; inst < TP16 / 8; inst += NB16) {
    const int tok = inst * 8 + (lane >> 3), cpos = lane & 7;
    const int chunk = cpos ^ img_key(tok);
    const void* g = tok < T16 ? (const void*)(src + (long long)tok * ld + chunk * 8)
                              : (const void*)(g_attn_zero_page + cpos * 16);
    glds16(g, dst + inst * 1024);
  }
}

// reversed, extended table of head h (natural units):  tabR[a] = table[M16 - a] for a <= M16, 0 for the two entries only
// padding slots reach, table[cls -> token] from CLSQ16 on
__device__ __forceinline__ void table16_setup(float* tabR, const float* table, int H, int h) {
  for (int a = threadIdx.x; a < TABLEN16; a += blockDim.x) {
    float v = 0.f;
    if (a <= M16) v = table[(long long)(M16 - a) * H + h];
    else if (a >= CLSQ16) v = table[(long long)(NRD16 - 3) * H + h];
    tabR[a] = v;
  }
}
// Forward: TWO copies of that table, the second shifted by one entry.  The four keys of a register group are consecutive
// entries; a pair starting at an EVEN entry of one of the copies is an aligned ds_read_b64 (2 LDS cycles for 2 values per
// lane; the ds_read2_b32 the compiler makes of unaligned pairs takes 4 -- the bias reads were 2/3 of the kernel's LDS time).
// Copy 1 holds entry a at index a + 1: a lane whose pair starts at an odd entry reads copy 1.
constexpr int TAB2LEN16 = 2 * TABLEN16 + 4;
__device__ __forceinline__ void table16_setup2(float* tabR, const float* table, int H, int h) {
  for (int a = threadIdx.x; a < TABLEN16; a += blockDim.x) {
    float v = 0.f;
    if (a <= M16) v = table[(long long)(M16 - a) * H + h];
    else if (a >= CLSQ16) v = table[(long long)(NRD16 - 3) * H + h];
    tabR[a] = v;
    tabR[TABLEN16 + 2 + a + 1] = v;
  }
}
// (as two separate instructions: the compiler would fuse neighbouring pairs into ds_read2_b64, 8 LDS cycles for 4 values)
// (volatile: the load/store optimizer would fuse neighbouring pairs into ds_read2_b64 -- 8 LDS cycles for 4 values)
__device__ __forceinline__ f32x2_t lds_f32x2_abs(unsigned lds_byte_addr) {
  return *reinterpret_cast<const volatile __attribute__((address_space(3))) f32x2_t*>(lds_byte_addr);
}
// per-lane index base of query q: a = base + keycode
__device__ __forceinline__ int q_base16(int q) {
  if (q == 0) return CLSQ16;
  if (q >= T16) return 0;
  const int u = q - 1;
  return M16 - ((u / W16) * R16 + (u % W16) + OFF16);
}
// byte offset of register (g, e) of key block 0 relative to the lane base (the lane base holds 4 hh; block kb adds 216 kb)
#define KOFF16(g, e) (4 * ((((g) >> 1) * R16) + 8 * ((g) & 1) + (e)))
constexpr int kBlockStep16 = 4 * 2 * R16;   // 216 bytes per key block

// ---- fragment addressing from TWO lane registers.  In the image layout of attn_common.hpp (128-byte rows, chunk c of token t
// at c ^ img_key(t)) the chunk and token indices of a lane's fragments differ by whole bits, so every fragment address is
// (lane base ^ constant) + wave-uniform offset -- one v_xad_u32 -- instead of a register per fragment (12 of them):
//   row fragment t (token r, chunk 2 t + hh):            (row0 ^ (t << 5)) + block
//   column fragment (ss, db), low / high half (j):       (col0 ^ (db << 6) ^ (j << 5)) + block + 2048 ss + 1024 j
//   (tokens + 8 flip bit 2 of (t >> 1), i.e. bit 1 of the key: the 32)
struct Lane16 { unsigned row0, col0; };
__device__ __forceinline__ Lane16 lane16(int lane) {
  const LaneOffs o = lane_offs(lane);
  return Lane16{(unsigned)o.row[0], (unsigned)o.col[0][0][0]};
}
// ... and the wave-uniform / compile-time parts of an address ride in the instruction's 16-bit offset field: a step computes
// 4 row bases + 2 column bases per image PAIR (V sits IMG16 bytes behind K, dO behind Q) instead of one address per read.
template <int OFF>
__device__ __forceinline__ bf16x8 lds_b128(unsigned addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(addr + OFF);
}
template <int OFF>
__device__ __forceinline__ s16x4 lds_tr16_off(unsigned lds_addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF) : "memory");
  return r;
}
struct RowBase16 { unsigned a[4]; };          // (row0 ^ (t << 5)) + block address, t = 0..3
struct ColBase16 { unsigned a[4]; };          // (col0 ^ (db << 6) ^ (j << 5)) + block address, index db + 2 j
__device__ __forceinline__ RowBase16 row_base16(const Lane16& l, unsigned blk) {
  RowBase16 b;
#pragma unroll
  for (int t = 0; t < 4; ++t) b.a[t] = (l.row0 ^ (t << 5)) + blk;
  return b;
}
__device__ __forceinline__ ColBase16 col_base16(const Lane16& l, unsigned blk) {
  return ColBase16{{l.col0 + blk, (l.col0 ^ 64u) + blk, (l.col0 ^ 32u) + blk, (l.col0 ^ 96u) + blk}};
}
template <int OFF, int SS, int DB>
__device__ __forceinline__ bf16x8 col_frag16(const ColBase16& b) {          // OFF: image displacement (0 or IMG16)
  union { struct { s16x4 l, h; } s; bf16x8 v; } u;
  u.s.l = lds_tr16_off<OFF + 2048 * SS>(b.a[DB]);
  u.s.h = lds_tr16_off<OFF + 2048 * SS + 1024>(b.a[DB + 2]);
  return u.v;
}

// Samples of one head are dealt to `nwg` workgroups: the first B % nwg take one more.  The workgroups with the smaller
// share have one sample period of slack: they start `stagger` cycles late (a per-workgroup fraction of it), so that the
// workgroups of the chip do not all run their memory phases (LDS-DMA of the next sample, stores of the last one) at the
// same moment -- in lockstep these are HBM bursts of ~45 MB that the whole chip then waits for.
struct Share16 { int b0, b1; bool slack; };
__device__ __forceinline__ Share16 share16(int j, int B, int nwg) {
  const int base = B / nwg, rem = B % nwg;
  Share16 s;
  s.b0 = j * base + (j < rem ? j : rem);
  s.b1 = s.b0 + base + (j < rem ? 1 : 0);
  s.slack = rem > 0 && j >= rem;
  return s;
}
__device__ __forceinline__ void stagger16(bool slack, int stagger) {
  if (!slack || stagger <= 0) return;
  const unsigned frac = ((blockIdx.x * 2654435761u) >> 22) & 1023u;        // 0 .. 1023
  const unsigned long long wait = ((unsigned long long)stagger * frac) >> 10;
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(16);
}
__device__ __forceinline__ void glds4(const void* gsrc, char* lds_dst) {     // LDS-DMA, 4 bytes per lane
  const unsigned lds = __builtin_amdgcn_readfirstlane(
      (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)lds_dst));
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(lds) : "memory", "m0");
}

// the same with a wave-uniform base and a 32-bit lane offset (no 64-bit per-lane pointer to keep alive)
__device__ __forceinline__ void glds4s(const void* sbase, unsigned voff, char* lds_dst) {
  const unsigned lds = __builtin_amdgcn_readfirstlane(
      (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)lds_dst));
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}

// 16 bytes per lane, wave-uniform base + 32-bit lane offset, LDS address given as a value
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
// The staging wave of the forward kernel keeps the lane offsets of all 28 instructions of a slot-ordered image in registers
// (it has nothing else to hold): an instruction is s_mov m0 + the load.  Padding slots read the cls row instead of the zero
// page (one base for all lanes): their scores get a bias of -inf and their probabilities are exactly 0, so any FINITE row
// serves.
struct SlotOffs16 { unsigned v[TP16 / 8]; };
__device__ __forceinline__ SlotOffs16 slot_offs16(long long ld) {
  const int lane = threadIdx.x & 63;
  SlotOffs16 o;
#pragma unroll
  for (int inst = 0; inst < TP16 / 8; ++inst) {
    const int slot = inst * 8 + (lane >> 3), cpos = lane & 7;
    const int tok = slot_tok(slot);
    o.v[inst] = (unsigned)(((tok >= 0 ? tok : 0) * (int)ld + (cpos ^ img_key(slot)) * 8) * 2);
  }
  return o;
}
__device__ __forceinline__ void stage_slots_fast(unsigned lds_dst, const __bf16* src, const SlotOffs16& o) {
#pragma unroll
  for (int inst = 0; inst < TP16 / 8; ++inst) glds16s(src, o.v[inst], lds_dst + inst * 1024);
}
// The backward kernel has no register to spare and no idle SIMD slot for an eighth wave to pay (measured: 299 - 330 us
// with a staging wave instead of 288, tools/exp/attn16_bwd_staging_wave.patch), so its seven waves stage their own share:
// instructions wave, wave + 7, wave + 14, wave + 21 of an image.  The general form above costs ~35 instructions per piece
// (64-bit multiplies, the zero-page select as EXEC-masked branches); here the grid row / token block of an instruction is
// wave-uniform (SCALAR base) and a lane's offset depends only on the PARITY of the instruction: two lane offsets,
// recomputed at every call from an opaque copy of the lane id (nothing stays live through the main loop), then
// s_mul / s_add + s_mov m0 + the load per piece.  Padding rows read a neighbouring REAL row (finite, see above).
__device__ __forceinline__ int img_key3(int k) { return img_key(2 * k); }     // key of the tokens 2 k, 2 k + 1 (k < 8)
__device__ __forceinline__ void stage_slots_lean(unsigned lds_dst, const __bf16* src, int ld, int wave) {
  int tl = (int)threadIdx.x;
  asm volatile("" : "+v"(tl));
  const int l3 = (tl >> 3) & 7, cpos = tl & 7;
  // even instruction: slots 16 ky + l3 (kx = l3); odd: kx = 8 + l3, slots 14 / 15 clamp to kx = 13
  const unsigned ve = (unsigned)(((1 + l3) * ld + (cpos ^ img_key3(l3 >> 1)) * 8) * 2);
  const int kxo = 8 + l3 < W16 ? 8 + l3 : W16 - 1;
  const unsigned co = (unsigned)((cpos ^ img_key3(4 + (l3 >> 1))) * 16);
  const unsigned vo = (unsigned)((1 + kxo) * ld * 2) + co;
  const unsigned vc = l3 == 6 ? co : vo;                   // instruction 1 holds slot 14 = the cls key: token 0
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int inst = wave + NB16 * k;                      // wave-uniform
    const __bf16* sb = src + (long long)((inst >> 1) * W16) * ld;
    const unsigned v = (inst & 1) ? (inst == 1 ? vc : vo) : ve;
    glds16s(sb, v, lds_dst + inst * 1024);
  }
}
// token order (Q, dO): tokens 8 inst + l3; rows >= 197 read token 192 + min(l3, 4)
__device__ __forceinline__ void stage_tokens_lean(unsigned lds_dst, const __bf16* src, int ld, int wave) {
  int tl = (int)threadIdx.x;
  asm volatile("" : "+v"(tl));
  const int l3 = (tl >> 3) & 7, cpos = tl & 7;
  const unsigned ce = (unsigned)((cpos ^ img_key3(l3 >> 1)) * 16), co = (unsigned)((cpos ^ img_key3(4 + (l3 >> 1))) * 16);
  const unsigned ve = (unsigned)(l3 * ld * 2) + ce, vo = (unsigned)(l3 * ld * 2) + co;
  const unsigned vt = (unsigned)((l3 < 4 ? l3 : 4) * ld * 2) + ce;            // instruction 24: tokens 192 .. 196 are real
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int inst = wave + NB16 * k;
    const bool tail = inst >= (T16 - 1) / 8;               // 24 and up
    const __bf16* sb = src + (long long)((tail ? (T16 - 1) / 8 : inst) * 8) * ld;
    const unsigned v = tail ? vt : ((inst & 1) ? vo : ve);
    glds16s(sb, v, lds_dst + inst * 1024);
  }
}

// -DATTN16_TIMING: per-section s_memtime totals of wave 0 / wave 4 of every workgroup (tools/attn16_sections.py)
#define T16_DECL()
#define T16_TICK(i)
#define T16_FLUSH()
// ------------------------------------------------------------------------------------------------ forward
// 8 waves: 7 compute a block of 32 queries each, the 8th only issues the LDS-DMA of the next sample.  Measured (B = 256,
// tools/exp/r04_run13.sh): with the K / V staging of the next sample left out the 7-wave kernel took 83 us instead of 109 --
// not memory time (the DMA has a whole sample period to land) but ISSUE time: a global_load_lds takes its wave 100+ cycles
// among busy neighbours, 8 per wave and sample, plus the slot -> token address arithmetic.  Wave 7 shares SIMD 3 with
// wave 3, the only SIMD of a 7-wave workgroup that held one wave.
constexpr int kThreadsFwd16 = kThreads16 + 64;
__global__ __launch_bounds__(kThreadsFwd16) void attn16_fwd_kernel(const __bf16* __restrict__ qkv, long long ldq, int B, int D,
                                                               int H, const float* __restrict__ table,
                                                               __bf16* __restrict__ out, long long ldo,
                                                               float* __restrict__ lse, int nwg, int stagger) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tabR = reinterpret_cast<float*>(smem);
  char* imgs = smem + TAB2LEN16 * 4;
  char* stg = imgs + 4 * IMG16;                // [7 waves][4 KiB] output staging
  const int h = blockIdx.x % H;
  const Share16 sh = share16(blockIdx.x / H, B, nwg);
  const int b0 = sh.b0, b1 = sh.b1;
  if (b0 >= b1) return;
  stagger16(sh.slack, stagger);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const bool hh1 = hh != 0;
  const unsigned sel_lo = sel_lo_reg();
  const Lane16 lo = lane16(lane);
  table16_setup2(tabR, table, H, h);
  const int q = wave * 32 + r;
  const int qc = q < T16 ? q : T16 - 1;
  // register groups of grid rows 2 kb (g = 0, 1) start at entry base + even, those of rows 2 kb + 1 (g = 2, 3) at base + odd
  unsigned bbe, bbo;
  {
    const int base = q_base16(q) + 4 * hh;
    const unsigned t0 = lds_addr_of(reinterpret_cast<const char*>(tabR));
    const unsigned c0 = t0 + 4 * base, c1 = t0 + 4 * (TABLEN16 + 2 + base + 1);
    bbe = (base & 1) ? c1 : c0;
    bbo = (base & 1) ? c0 : c1;
  }
  const float clsb = table[(long long)(q == 0 ? NRD16 - 1 : NRD16 - 2) * H + h];
  {
    const __bf16* s0 = qkv + (long long)b0 * T16 * ldq + h * HD;
    stage_slots(imgs, s0 + D, ldq, wave, NB16 + 1);
    stage_slots(imgs + IMG16, s0 + 2 * D, ldq, wave, NB16 + 1);
  }
  if (wave == NB16) {                          // the staging wave
    const SlotOffs16 so = slot_offs16(ldq);
    // barrier j = 2 b: the early waves start the scores of sample b (K image), the late ones the softmax / PV of b - 1 (V);
    // j = 2 b + 1: the other way round.  K(b - 1) is dead at barrier 2 b, V(b - 1) at 2 b + 1: K(b + 1) / V(b + 1) are
    // issued there and have a whole period (two barriers) to land; the piece issued after barrier j - 1 may still be in
    // flight at barrier j (vector-memory operations complete in order: vmcnt(28) = all but the newest image).
    const int n = b1 - b0;
    bool pend = false;
    for (int j = 0; j <= 2 * n; ++j) {
      if (pend) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const int b = j >> 1;
      pend = b + 1 < n;
      if (pend) {
        const __bf16* s1 = qkv + (long long)(b0 + b + 1) * T16 * ldq + h * HD;
        const unsigned dst = lds_addr_of(imgs + ((b + 1) & 1) * 2 * IMG16);
        if (j & 1) stage_slots_fast(dst + IMG16, s1 + 2 * D, so);
        else stage_slots_fast(dst, s1 + D, so);
      }
    }
    return;
  }
  bf16x8 Qn[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) Qn[t] = ld16(qkv + ((long long)b0 * T16 + qc) * ldq + h * HD + 16 * t + 8 * hh);
  T16_DECL();
  // waves 4..6 (the second wave of SIMD 0..2) run HALF A SAMPLE behind waves 0..3: while one wave of a SIMD is in its MFMA
  // phases the other is in its VALU phases (in lockstep the two add up: 28 + 28 MFMAs with the vector ALU idle, then
  // bias / max / exp with the matrix core idle).  Same code, one barrier more in front (late) or behind (early).
  const bool late = __builtin_amdgcn_readfirstlane(wave) >= 4;
  if (late) { ATTN16_WAIT_VM(4); __syncthreads(); }
  for (int b = b0; b < b1; ++b) {
    T16_TICK(5);
    const int cur = (b - b0) & 1;
    const char* Ks = imgs + cur * 2 * IMG16;
    const char* Vs = Ks + IMG16;
    bf16x8 Qf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) Qf[t] = Qn[t];
    // vector-memory operations complete in issue order: the Q rows of this sample (and, the first time, this wave's
    // share of the first images) have landed once at most the 4 output stores of the previous sample (issued after them,
    // every wave issues all four: padding rows go to the trash page) are still in flight.  The count must never exceed the
    // stores really issued behind the loads, so the lse store (predicated; the oldest of the five) is not counted.
    ATTN16_WAIT_VM(4);
    __syncthreads();                         // sample b's images (and, the first time, the table) landed; b-1 consumed
    T16_TICK(0);
    if (b + 1 < b1) {
      const __bf16* s1 = qkv + (long long)(b + 1) * T16 * ldq + h * HD;
#pragma unroll
      for (int t = 0; t < 4; ++t) Qn[t] = ld16(s1 + (long long)qc * ldq + 16 * t + 8 * hh);
    }
    const RowBase16 kr = row_base16(lo, lds_addr_of(Ks));
    const ColBase16 vc = col_base16(lo, lds_addr_of(Vs));
    f32x16 s[NB16];
#pragma unroll
    for (int kb = 0; kb < NB16; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) s[kb] = MFMA32(lds_b128<0>(kr.a[t] + kb * 4096), Qf[t], s[kb]);
    }
    T16_TICK(1);
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NB16; ++kb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float bias[4];
        {
          const unsigned a0 = ((g >> 1) ? bbo : bbe) + kb * kBlockStep16 + KOFF16(g, 0);
          const f32x2_t b01 = lds_f32x2_abs(a0), b23 = lds_f32x2_abs(a0 + 8);
          bias[0] = b01[0]; bias[1] = b01[1]; bias[2] = b23[0]; bias[3] = b23[1];
        }
        if (g & 1) {                           // slots 14, 15 of a grid row: padding, except the cls key (block 0)
          bias[2] = hh1 ? (kb == 0 && g == 1 ? clsb : -INFINITY) : bias[2];
          bias[3] = hh1 ? -INFINITY : bias[3];
        }
        const unsigned p0 = pk_bf16(s[kb][4 * g], s[kb][4 * g + 1]), p1 = pk_bf16(s[kb][4 * g + 2], s[kb][4 * g + 3]);
        s[kb][4 * g] = add_lo(p0, bias[0], sel_lo);
        s[kb][4 * g + 1] = add_hi(p0, bias[1]);
        s[kb][4 * g + 2] = add_lo(p1, bias[2], sel_lo);
        s[kb][4 * g + 3] = add_hi(p1, bias[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[kb][4 * g + e]);
      }
    }
    T16_TICK(2);
    __syncthreads();                         // the V image of sample b has landed
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mneg = -mx * kLog2e;
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB16; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p = fexp2(fmaf(s[kb][i], kLog2e, mneg));
        s[kb][i] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    if (hh == 0 && q < T16) lse[((long long)b * H + h) * TP16 + q] = mx + flog2(sum) * kLn2;
    T16_TICK(3);
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB16; ++kb) {
      bf16x8 vf[2][2];
      {
        const ColBase16 vb{{vc.a[0] + kb * 4096, vc.a[1] + kb * 4096, vc.a[2] + kb * 4096, vc.a[3] + kb * 4096}};   // (constants: fold)
        vf[0][0] = col_frag16<0, 0, 0>(vb); vf[0][1] = col_frag16<0, 0, 1>(vb);
        vf[1][0] = col_frag16<0, 1, 0>(vb); vf[1][1] = col_frag16<0, 1, 1>(vb);
      }
      bf16x8 pf[2];
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) pf[ss] = acc_frag(s[kb], ss, inv);
      LDS_TR_WAIT();
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int db = 0; db < 2; ++db) o[db] = MFMA32(vf[ss][db], pf[ss], o[db]);
    }
    T16_TICK(4);
    {
      char* st = stg + wave * 4096;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (__bf16)o[db][4 * g + e];
          tile_put(st, r, hh, db, g, w);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (lane >> 3) + 8 * i, qq = wave * 32 + row;
        const bf16x8 v = tile_get(st, row, lane & 7);
        __bf16* dst = qq < T16 ? out + ((long long)b * T16 + qq) * ldo + h * HD + (lane & 7) * 8
                               : reinterpret_cast<__bf16*>(g_attn16_trash) + lane * 8;
        *reinterpret_cast<bf16x8*>(dst) = v;
      }
    }
  }
  if (!late) __syncthreads();
  T16_TICK(5);
}

// ------------------------------------------------------------------------------------------------ fused backward
// LDS map (bytes)
constexpr int kTabBytes16 = TABLEN16 * 4;                 // 4400
constexpr int kBinsI16 = kTabBytes16;                     // int32 fixed-point buckets of the current sample
constexpr int kBinsF16 = 2 * kTabBytes16;                 // fp32 buckets of the workgroup
constexpr int kRows16 = 3 * kTabBytes16;                  // [3][256] floats: lse, delta, |dO|^2 of the sample's queries
constexpr int kRowsLd16 = 256;                            // (four wave-instructions of 64 lanes each)
constexpr int kRed16 = kRows16 + 3 * kRowsLd16 * 4;                   // [8][4] floats: per-wave maxima of the bound terms
constexpr int kQsum16 = kRed16 + 128;                     // [64] floats
constexpr int kImgs16 = (kQsum16 + 256 + 15) & ~15;       // Q dO K V
constexpr int kExch16 = kImgs16 + 4 * IMG16;              // [7 waves][P^T tile | dS^T tile], 2 KiB each
constexpr int kLdsBwd16 = kExch16 + NB16 * 4096;
static_assert(kLdsBwd16 <= 160 * 1024, "fused backward: LDS budget");

// max over the wave by DPP (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31: no LDS traffic, 6 VALU), result uniform (lane 63)
__device__ __forceinline__ float wave_max(float v) {
#define A16_DPP_MAX(ctrl, rmask)                                                                         \
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rmask, 0xf, false)))
  A16_DPP_MAX(0x111, 0xf);
  A16_DPP_MAX(0x112, 0xf);
  A16_DPP_MAX(0x114, 0xf);
  A16_DPP_MAX(0x118, 0xf);
  A16_DPP_MAX(0x142, 0xa);
  A16_DPP_MAX(0x143, 0xc);
#undef A16_DPP_MAX
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// value of lane - 1 / lane + 1 of the WAVE (0 beyond its ends): v_mov_b32_dpp wave_shr:1 / wave_shl:1 -- the GFX9 whole-wave
// shifts exist on gfx950 and cross the 16-lane rows (tools/micro/dpp_wave.hip)
__device__ __forceinline__ float dpp_shr1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shl1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

// exchange tile [32 queries][32 keys] bf16, 64-byte rows, 8-byte unit u of row r at position u ^ ((r >> 1) & 7): the
// ds_write_b64 of 16 consecutive rows and the transposing reads (32 lanes = 4 rows x 8 units) are both conflict-free
__device__ __forceinline__ int exch_off(int row, int unit) { return row * 64 + ((unit ^ ((row >> 1) & 7)) << 3); }

// FD: delta = rowsum(dO * O) and |dO|^2 of a wave's queries are computed HERE from the forward output `oimg` (the rows of
// its 32 queries, loaded a sample ahead into 16 registers by inline-asm loads that sit in front of the previous sample's
// stores in the in-order queue) and the dO rows it reads from LDS anyway: 32 v_dot2_f32_bf16 per wave and sample instead of
// the separate attn_delta pass over dO and O (25 us per layer at B = 256) and two of the three per-query row DMAs.
template <bool DT, bool FD>
__global__ __launch_bounds__(kThreads16) void attn16_bwd_kernel(const __bf16* __restrict__ qkv, long long ldq,
                                                               const __bf16* __restrict__ dout, long long ldo,
                                                               const __bf16* __restrict__ oimg, long long ldoo,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ delta,
                                                               const float* __restrict__ table,
                                                               __bf16* __restrict__ dqkv, long long lddq,
                                                               float* __restrict__ dtable, float* __restrict__ dqbias,
                                                               int B, int D, int H, float scale, int nwg, int stagger) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tabR = reinterpret_cast<float*>(smem);
  int* binsi = reinterpret_cast<int*>(smem + kBinsI16);
  float* rowsL = reinterpret_cast<float*>(smem + kRows16);
  float* binsf = reinterpret_cast<float*>(smem + kBinsF16);
  float* red = reinterpret_cast<float*>(smem + kRed16);
  float* qsum = reinterpret_cast<float*>(smem + kQsum16);
  char* Qs = smem + kImgs16;
  char* dOs = Qs + IMG16;
  char* Ks = dOs + IMG16;
  char* Vs = Ks + IMG16;
  char* exch = smem + kExch16;
  const int h = blockIdx.x % H;
  const Share16 sh = share16(blockIdx.x / H, B, nwg);
  const int b0 = sh.b0, b1 = sh.b1;
  if (b0 >= b1) return;
  stagger16(sh.slack, stagger);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const bool hh1 = hh != 0;
  const unsigned sel_lo = sel_lo_reg();
  Lane16 lo = lane16(lane);
  table16_setup(tabR, table, H, h);
  for (int i = threadIdx.x; i < TABLEN16; i += blockDim.x) { binsi[i] = 0; binsf[i] = 0.f; }
  if (threadIdx.x < HD) qsum[threadIdx.x] = 0.f;
  const int q = wave * 32 + r;
  const unsigned bb = lds_addr_of(reinterpret_cast<const char*>(tabR)) + 4 * (q_base16(q) + 4 * hh);
  const float clsb = table[(long long)(q == 0 ? NRD16 - 1 : NRD16 - 2) * H + h];
  // exchange tile addressing: writer (row = r, units 2 g + hh), reader (transposing 8-byte reads, see attn_common.hpp)
  // (the unit and row indices of a lane's accesses differ by whole bits again: one register each, XOR constants)
  unsigned wr0 = exch_off(r, hh);                      // unit 2 g + hh: wr0 ^ (g << 4)
  unsigned rd0 = exch_off(4 * hh + ((lane >> 2) & 3), ((lane >> 4) & 1) * 4 + (lane & 3));
                                                             // rows + 16 ss + 8 j: (rd0 ^ (j << 5)) + 1024 ss + 512 j
  float bsum[8];                                             // q_bias gradient: columns 8 (lane & 7) .. + 7 of the rows this lane stores
#pragma unroll
  for (int i = 0; i < 8; ++i) bsum[i] = 0.f;
  // diagonal chains of the bucket sums (see the produce step)
  // RUNS of linked lanes: lane r of a half-wave is linked to lane r - 1 when query q - 1 sits in the same grid row (padding queries
  // -- all their terms are exactly 0 -- form one run of their own).  A run's last lane (`chain_end`) is left with three unfinished
  // chains (keys 0, 0..1, 0..2 of a register group); their buckets are bucket(lane; key e) = bucket(lane - e; key 0) inside a
  // run, so lanes t, t - 1, t - 2 of the run add them -- at their OWN key-0 address, in ONE wave-wide atomic (fEnd / fD1 / fD2
  // select a lane's term; every other lane adds 0).  Only runs shorter than three lanes (the cls query; query 127, alone between
  // a grid-row boundary and the end of wave 3) still need the separate adds s1 / s2: wave-uniform branches, taken by two
  // of the seven waves.  Per register group: 2 atomic instructions instead of 4 (an LDS atomic costs its wave ~52 cycles
  // whatever its lane count).
  auto has_left_of = [&](int rr, int qq) {
    if (rr <= 0 || rr > 31) return false;
    return qq >= T16 ? qq - 1 >= T16 : (qq >= 2 && (qq - 1) % W16 != 0);
  };
  auto is_end_of = [&](int rr, int qq) { return rr <= 31 && !has_left_of(rr + 1, qq + 1); };
  const bool has_left = has_left_of(r, q);
  const float link = has_left ? 1.f : 0.f;
  const bool chain_end = is_end_of(r, q);
  const float fEnd = chain_end ? 1.f : 0.f;
  // (the neighbours' flags through the same whole-wave shifts the chains use: lanes 31 / 32 never link, so nothing crosses halves)
  const float e1 = fEnd * link;                                  // a run end with a lane to its left
  const float e2 = e1 * dpp_shr1(link);                          // ... and one more
  const float fD1 = dpp_shl1(e1);                                // lane + 1 is such an end: its chain 1 is added here
  const float fD2 = dpp_shl1(dpp_shl1(e2));                      // lane + 2 is: its chain 2 is added here
  const bool s1 = chain_end && !has_left;                        // run of one lane: chains 1 and 2 stay with it
  const bool s2 = chain_end && e2 == 0.f;                        // run of one or two lanes: chain 2 stays with its end
  const float fS1 = s1 ? 1.f : 0.f, fS2 = s2 ? 1.f : 0.f;
  const bool any_s1 = __builtin_amdgcn_ballot_w64(s1) != 0, any_s2 = __builtin_amdgcn_ballot_w64(s2) != 0;
  float dcls = 0.f;                                          // gradient of the (token -> cls) / (cls -> cls) bucket
  // per-query rows of the sample (lse, delta, |dO|^2) travel by LDS-DMA as well: a plain load into registers would make the
  // compiler wait for ALL vector-memory operations (the stores of the previous sample included) in front of the first use
  auto stage_rows = [&](int b) {
    if (FD) {                                                // only the lse row; padding lanes read lse row entries < TP16 too
      if (wave < 4) {
        int tl = (int)threadIdx.x;
        asm volatile("" : "+v"(tl));                         // (recomputed per call: see load_o)
        const unsigned voff = (unsigned)((tl < TP16 ? tl : TP16 - 1) * 4);
        glds4s(lse + ((long long)b * H + h) * TP16, voff, reinterpret_cast<char*>(rowsL) + wave * 256);
      }
      return;
    }
    if (wave < 4) {
      const int t = wave * 64 + lane;
      const bool ok = t < T16;
      const void* z = g_attn_zero_page + (lane & 31) * 4;
      const long long row = (long long)b * T16 + t;
      glds4(ok ? (const void*)(lse + ((long long)b * H + h) * TP16 + t) : z, reinterpret_cast<char*>(rowsL) + wave * 256);
      if (!FD) {
        glds4(ok ? (const void*)(delta + row * H + h) : z, reinterpret_cast<char*>(rowsL + kRowsLd16) + wave * 256);
        glds4(ok ? (const void*)(delta + ((long long)B * T16 + row) * H + h) : z, reinterpret_cast<char*>(rowsL + 2 * kRowsLd16) + wave * 256);
      }
    }
  };
  // forward-output rows of this lane's query (dims 16 t + 8 hh .. + 7, t = 0..3: the layout of the row fragments), as asm
  // loads: hipcc does not see them, so it neither waits for them nor for anything else in front of their first use; they
  // are issued with the sample's LDS-DMA, i.e. in FRONT of the previous sample's stores, and have landed when the counted
  // wait at the top of the sample has passed
  // (scalar sample base + a 32-bit lane offset that is RECOMPUTED at every call from an opaque copy of the lane id: hoisted
  // out of the sample loop it would be one more register that is live through the main loop, which has none to spare)
  bf16x8 On[4];
  auto load_o = [&](int b) {
    int ql = (int)threadIdx.x;
    asm volatile("" : "+v"(ql));
    const int qq = (ql >> 6) * 32 + (ql & 31);
    const int qcl = qq < T16 ? qq : T16 - 1;
    const unsigned voff = (unsigned)((qcl * (int)ldoo + h * HD + 8 * ((ql >> 5) & 1)) * 2);     // bytes; rows * ld * 2 < 2^31
    const __bf16* sb = oimg + (long long)b * T16 * ldoo;                                          // wave-uniform
    asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:32\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:64\n\tglobal_load_dwordx4 %3, %4, %5 offset:96"
                 : "=&v"(On[0]), "=&v"(On[1]), "=&v"(On[2]), "=&v"(On[3]) : "v"(voff), "s"(sb) : "memory");
  };
  auto stage_sample = [&](int b, bool kv) {                 // kv = false: K / V were staged early (last step of the sample before)
    const __bf16* s = qkv + (long long)b * T16 * ldq + h * HD;
    stage_tokens(Qs, s, ldq);
    stage_tokens(dOs, dout + (long long)b * T16 * ldo + h * HD, ldo);
    if (kv) {
      stage_slots(Ks, s + D, ldq);
      stage_slots(Vs, s + 2 * D, ldq);
    }
  };
  stage_rows(b0);
  stage_sample(b0, true);
  if (FD) load_o(b0);
  // the first sample has no stores behind its DMA for the counted wait below to leave in flight: everything lands here
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  T16_DECL();
  for (int b = b0; b < b1; ++b) {
    T16_TICK(9);
    // in issue order: [LDS-DMA of this sample: rows, images] [12 stores of the previous sample]: the stores may stay in
    // flight under this sample's compute.  Two of the twelve are left out of the count as a margin (the count must never
    // exceed the stores really issued behind the DMA; waiting for the two oldest costs nothing measurable).
    ATTN16_WAIT_VM(10);
    __syncthreads();                                         // images of sample b (first time: tables) are in LDS
    const float lq2 = q < T16 ? rowsL[q] * kLog2e : INFINITY;   // padding queries: p = exp2(-inf) = 0
    float dq_, nqn;
    if constexpr (FD) {
      asm volatile("" : "+v"(On[0]), "+v"(On[1]), "+v"(On[2]), "+v"(On[3]));      // landed (see load_o); named first here
      const RowBase16 qr0 = row_base16(lo, lds_addr_of(Qs) + wave * 4096);
      float dh = 0.f, nh = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        union { bf16x8 v; bf16x2_t p[4]; } d, o;
        d.v = lds_b128<IMG16>(qr0.a[t]);                   // dO row of this lane's query (zero rows for padding queries)
        o.v = On[t];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dh = __builtin_amdgcn_fdot2_f32_bf16(d.p[i], o.p[i], dh, false);
          nh = __builtin_amdgcn_fdot2_f32_bf16(d.p[i], d.p[i], nh, false);
        }
      }
      dq_ = dh + __shfl_xor(dh, 32);                       // the other half of the head dimension sits 32 lanes away
      nqn = nh + __shfl_xor(nh, 32);
    } else {
      dq_ = rowsL[kRowsLd16 + q];
      nqn = rowsL[2 * kRowsLd16 + q];
    }
    T16_TICK(0);
    // ---- fixed-point scale of the bucket atomics: |dS| = p |dP - delta| <= max|dO_q| max|V_k| + max|delta_q| =: bound;
    // 2^19 / bound: fx_round() needs |x| < 2^22 and a chain sum has up to four terms (bf16-rounded dP may pass the bound by
    // 2^-8); a bucket's <= 196 terms stay far below 2^31
    float fx = 0.f;
    if (DT) {
      float vn = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 v = lds_b128<0>((lo.row0 ^ (t << 5)) + lds_addr_of(Vs) + wave * 4096);   // V rows of this wave's key slots
#pragma unroll
        for (int i = 0; i < 8; ++i) vn = fmaf((float)v[i], (float)v[i], vn);
      }
      // (padding slots carry a copy of a real key row: only half of a padding lane's |V|^2 can be missing, a bound stays a bound)
      const float m0 = wave_max(nqn), m1 = wave_max(fabsf(dq_)), m2 = 2.0f * wave_max(vn);   // vn: half a row per lane
      if (lane == 0) { red[wave * 4 + 0] = m0; red[wave * 4 + 1] = m1; red[wave * 4 + 2] = m2; }
      __syncthreads();
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < NB16; ++w) {
        t0 = fmaxf(t0, red[w * 4 + 0]);
        t1 = fmaxf(t1, red[w * 4 + 1]);
        t2 = fmaxf(t2, red[w * 4 + 2]);
      }
      const float bound = sqrtf(t0) * sqrtf(t2) + t1;
      fx = bound > 0.f ? 524288.0f / bound : 0.f;
    }
    f32x16 dQt[2], dKt[2], dVt[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) { dQt[db][i] = 0.f; dKt[db][i] = 0.f; dVt[db][i] = 0.f; }
    T16_TICK(1);
    // ---------------- consume: tile (queries wpv, keys `wave`)
    auto consume = [&](int wpv) {
      const unsigned tb = lds_addr_of(exch) + wpv * 4096;
      const ColBase16 qc = col_base16(lo, lds_addr_of(Qs) + wpv * 4096);      // dO image: + IMG16
      const unsigned rd[2] = {rd0 + tb, (rd0 ^ 32u) + tb};                     // exchange rows + 8 j
      auto half = [&](auto SS) {
        constexpr int ss = decltype(SS)::value;
        union { struct { s16x4 l, h; } s; bf16x8 v; } pB, dB;
        pB.s.l = lds_tr16_off<1024 * ss>(rd[0]);
        pB.s.h = lds_tr16_off<1024 * ss + 512>(rd[1]);
        dB.s.l = lds_tr16_off<2048 + 1024 * ss>(rd[0]);
        dB.s.h = lds_tr16_off<2048 + 1024 * ss + 512>(rd[1]);
        bf16x8 cdo[2], cq[2];
        cdo[0] = col_frag16<IMG16, ss, 0>(qc); cdo[1] = col_frag16<IMG16, ss, 1>(qc);
        cq[0] = col_frag16<0, ss, 0>(qc); cq[1] = col_frag16<0, ss, 1>(qc);
        LDS_TR_WAIT();
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dVt[db] = MFMA32(cdo[db], pB.v, dVt[db]);
          dKt[db] = MFMA32(cq[db], dB.v, dKt[db]);
        }
      };
      half(std::integral_constant<int, 0>{});
      half(std::integral_constant<int, 1>{});
    };
    int kb = wave, wp = wave;                                // key block produced / producer consumed in this step
    for (int s = 0; s < NB16; ++s) {
      // (keeps the XOR forms of the fragment addresses inside the loop: hoisted, they occupy a register each)
      asm volatile("" : "+v"(lo.row0), "+v"(lo.col0), "+v"(wr0), "+v"(rd0));
      // ---------------- produce: tile (queries `wave`, keys kb)
      const unsigned kblk = lds_addr_of(Ks) + kb * 4096;
      const RowBase16 kr = row_base16(lo, kblk);               // K rows of block kb; V rows: + IMG16
      const RowBase16 qr = row_base16(lo, lds_addr_of(Qs) + wave * 4096);   // this wave's Q rows; dO rows: + IMG16
      f32x16 S, dP;
#pragma unroll
      for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        S = MFMA32(lds_b128<0>(kr.a[t]), lds_b128<0>(qr.a[t]), S);
        dP = MFMA32(lds_b128<IMG16>(kr.a[t]), lds_b128<IMG16>(qr.a[t]), dP);
      }
      T16_TICK(2);
      const int ba = (int)bb + kb * kBlockStep16;
      const int na = ba + kBinsI16;
      const float padb = kb == 0 ? clsb : -INFINITY;
      // the bias reads of register group g + 1 are issued BEFORE the bucket atomics of group g: LDS operations retire in
      // order, so a read behind an atomic waits for it (measured: the four exposed atomic latencies per tile cost more than
      // all of the softmax arithmetic)
      float bnext[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) bnext[e] = lds_f32_abs(ba + KOFF16(0, e));
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float bias[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bias[e] = bnext[e];
        if (g < 3) {
#pragma unroll
          for (int e = 0; e < 4; ++e) bnext[e] = lds_f32_abs(ba + KOFF16((g + 1) & 3, e));
        }
        if (g & 1) {
          bias[2] = hh1 ? (g == 1 ? padb : -INFINITY) : bias[2];
          bias[3] = hh1 ? -INFINITY : bias[3];
        }
        const unsigned s0 = pk_bf16(S[4 * g], S[4 * g + 1]), s1 = pk_bf16(S[4 * g + 2], S[4 * g + 3]);
        const unsigned d0 = pk_bf16(dP[4 * g], dP[4 * g + 1]), d1 = pk_bf16(dP[4 * g + 2], dP[4 * g + 3]);
        float tt[4], dd[4];
        tt[0] = add_lo(s0, bias[0], sel_lo); tt[1] = add_hi(s0, bias[1]); tt[2] = add_lo(s1, bias[2], sel_lo); tt[3] = add_hi(s1, bias[3]);
        dd[0] = add_lo(d0, -dq_, sel_lo); dd[1] = add_hi(d0, -dq_); dd[2] = add_lo(d1, -dq_, sel_lo); dd[3] = add_hi(d1, -dq_);
        float xb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = fexp2(fmaf(tt[e], kLog2e, -lq2));
          const float ds = p * dd[e];
          S[4 * g + e] = p;
          dP[4 * g + e] = ds;
          xb[e] = ds;
        }
        if (DT) {
          if (g == 1 && kb == 0) {                           // the cls key: its bucket is kept in a register
            dcls += hh1 ? xb[2] : 0.f;
            xb[2] = hh1 ? 0.f : xb[2];
          }
          // (query + 1, key + 1) is the bucket of (query, key): the four keys of a register group are summed along the
          // diagonal over neighbouring lanes first (wave_shr:1; `link` = 0 where the left neighbour is in another grid row
          // or the other half-wave), so a lane adds ONE chain sum instead of four elements; the three unfinished chains of a
          // run's last lane go out in one more wave-wide atomic (see the run flags above).
          const float c1 = fmaf(dpp_shr1(xb[0]), link, xb[1]);
          const float c2 = fmaf(dpp_shr1(c1), link, xb[2]);
          const float c3 = fmaf(dpp_shr1(c2), link, xb[3]);
          lds_add_i32_abs(na + KOFF16(g, 3), fx_round(c3, fx));
          {
            const float wsum = fmaf(fD2, dpp_shl1(dpp_shl1(c2)), fmaf(fD1, dpp_shl1(c1), fEnd * xb[0]));
            lds_add_i32_abs(na + KOFF16(g, 0), fx_round(wsum, fx));
          }
          // (wave-uniform branches; the other lanes of such a wave add 0 -- a per-lane `if (s1)` here was compiled by hipcc
          // (ROCm 7.2) into a test of an unrelated data register: tools/dt_dbg.py)
          if (any_s1) lds_add_i32_abs(na + KOFF16(g, 1), fx_round(fS1 * c1, fx));
          if (any_s2) lds_add_i32_abs(na + KOFF16(g, 2), fx_round(fS2 * c2, fx));
        }
      }
      const bf16x8 pf0 = acc_frag(S, 0, 1.0f), pf1 = acc_frag(S, 1, 1.0f);
      const bf16x8 df0 = acc_frag(dP, 0, 1.0f), df1 = acc_frag(dP, 1, 1.0f);
      T16_TICK(3);
      {
        bf16x8 ckf[2][2];
        const ColBase16 kc = col_base16(lo, kblk);
        ckf[0][0] = col_frag16<0, 0, 0>(kc); ckf[0][1] = col_frag16<0, 0, 1>(kc);
        ckf[1][0] = col_frag16<0, 1, 0>(kc); ckf[1][1] = col_frag16<0, 1, 1>(kc);
        LDS_TR_WAIT();
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dQt[db] = MFMA32(ckf[0][db], df0, dQt[db]);
          dQt[db] = MFMA32(ckf[1][db], df1, dQt[db]);
        }
      }
      T16_TICK(4);
      __syncthreads();                                       // every wave has consumed the tiles of step s - 1
      T16_TICK(5);
      if (s == NB16 - 1 && b + 1 < b1) {                     // ... and finished its last produce: the K / V images are dead
        const __bf16* sn = qkv + (long long)(b + 1) * T16 * ldq + h * HD;
        stage_slots_lean(lds_addr_of(Ks), sn + D, (int)ldq, wave);
        stage_slots_lean(lds_addr_of(Vs), sn + 2 * D, (int)ldq, wave);
      }
      {
        const unsigned mine = lds_addr_of(exch) + wave * 4096;
        unsigned wa[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) wa[g] = (wr0 ^ (g << 4)) + mine;
        typedef __attribute__((address_space(3))) unsigned long long lds_u64;
        auto put = [&](int off, const bf16x8& v, int g0) {
          union { bf16x8 v; unsigned long long h[2]; } u;
          u.v = v;
          *reinterpret_cast<lds_u64*>(wa[g0] + off) = u.h[0];
          *reinterpret_cast<lds_u64*>(wa[g0 + 1] + off) = u.h[1];
        };
        put(0, pf0, 0); put(0, pf1, 2); put(2048, df0, 0); put(2048, df1, 2);
      }
      __syncthreads();                                       // tiles of step s are published
      T16_TICK(6);
      consume(wp);
      kb = kb + 1 == NB16 ? 0 : kb + 1;
      wp = wp == 0 ? NB16 - 1 : wp - 1;
      T16_TICK(7);
    }
    __syncthreads();                                         // all reads of the images and all bucket atomics are done
    T16_TICK(8);
    if constexpr (FD) load_o(b + 1 < b1 ? b + 1 : b);      // in front of the Q / dO DMA of the next sample
    if (b + 1 < b1) {
      stage_rows(b + 1);
      stage_tokens_lean(lds_addr_of(Qs), qkv + (long long)(b + 1) * T16 * ldq + h * HD, (int)ldq, wave);
      stage_tokens_lean(lds_addr_of(dOs), dout + (long long)(b + 1) * T16 * ldo + h * HD, (int)ldo, wave);
    }
    // ---------------- epilogue of sample b (under the LDS-DMA of sample b + 1)
    if (DT) {
      const float inv = fx > 0.f ? 1.0f / fx : 0.f;
      for (int i = threadIdx.x; i < TABLEN16; i += blockDim.x) {
        const int v = binsi[i];
        if (v != 0) { binsf[i] += (float)v * inv; binsi[i] = 0; }
      }
    }
    {
      char* st = exch + wave * 4096;                       // this wave's exchange slot is free until step 0 of the next sample
      const int c8 = (lane & 7) * 8;
      // dQ:  d(q_lin) = d(q') * scale
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (__bf16)(bfr(dQt[db][4 * g + e]) * scale);
          tile_put(st, r, hh, db, g, w);
        }
      // (the forward-output rows of the next sample are requested HERE: the dQ accumulators are dead -- 16 registers are
      // free -- and none of the sample's twelve stores has been issued yet, so the loads wait for nothing but the DMA in
      // front of them and all twelve stores stay behind them for the counted wait at the top of the next sample)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (lane >> 3) + 8 * i, qq = wave * 32 + row;
        const bf16x8 v = tile_get(st, row, lane & 7);
        __bf16* dst = qq < T16 ? dqkv + ((long long)b * T16 + qq) * lddq + h * HD + c8
                               : reinterpret_cast<__bf16*>(g_attn16_trash) + lane * 8;
        *reinterpret_cast<bf16x8*>(dst) = v;
        if (qq < T16) {
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[j] += (float)v[j];
        }
      }
      // dK, dV: rows are the tokens of this wave's key slots
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (__bf16)(m == 0 ? dKt[db][4 * g + e] : dVt[db][4 * g + e]);
            tile_put(st, r, hh, db, g, w);
          }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = (lane >> 3) + 8 * i, tok = slot_tok(wave * 32 + row);
          const bf16x8 v = tile_get(st, row, lane & 7);
          __bf16* dst = tok >= 0 ? dqkv + ((long long)b * T16 + tok) * lddq + (m + 1) * D + h * HD + c8
                                 : reinterpret_cast<__bf16*>(g_attn16_trash) + lane * 8;
          *reinterpret_cast<bf16x8*>(dst) = v;
        }
      }
    }
  }
  T16_TICK(9);
  T16_FLUSH();
  __syncthreads();
  if (DT) {
    for (int a = threadIdx.x; a <= M16; a += blockDim.x) {
      const float v = binsf[a];
      if (v != 0.f) atomicAdd(dtable + (long long)(M16 - a) * H + h, v);
    }
    if (wave == 0) {                                         // the cls query's constant region -> bucket (cls -> token)
      float v = 0.f;
      for (int a = CLSQ16 + lane; a < TABLEN16; a += 64) v += binsf[a];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) atomicAdd(dtable + (long long)(NRD16 - 3) * H + h, v);
    }
    // cls key column: (cls, cls) from the cls query's lane, (token -> cls) from everybody else
    const float own = q == 0 ? dcls : 0.f;
    float oth = q == 0 ? 0.f : dcls;
    for (int o = 32; o > 0; o >>= 1) oth += __shfl_xor(oth, o);
    if (lane == 0) atomicAdd(dtable + (long long)(NRD16 - 2) * H + h, oth);
    if (q == 0 && hh1) atomicAdd(dtable + (long long)(NRD16 - 1) * H + h, own);
  }
  if (dqbias) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = bsum[j];
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane < 8) atomicAdd(qsum + lane * 8 + j, v);
    }
    __syncthreads();
    if (threadIdx.x < HD) atomicAdd(dqbias + h * HD + threadIdx.x, qsum[threadIdx.x]);
  }
}

int num_cu16(hipStream_t s) { const int n = memhip::usable_cus(s); return n > 0 ? n : 256; }
// workgroups per head: one workgroup per CU (LDS), as many as fit in one round
int nwg16(int B, int heads, hipStream_t s) {
  const int n = num_cu16(s) / heads;
  return n < 1 ? 1 : (n > B ? B : n);
}

}  // namespace

namespace memhip {

bool attn16_fits(int T, int window_h, int window_w) { return window_h == W16 && window_w == W16 && T == T16; }

int attn16_fwd(const void* qkv, int64_t ldqkv, int B, int D, int heads, const float* table, void* out, int64_t ldo,
               float* lse, hipStream_t s) {
  const size_t sm = (size_t)TAB2LEN16 * 4 + 4 * IMG16 + NB16 * 4096;
  static bool attr_done = false;
  if (!attr_done) {
    MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn16_fwd_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
    attr_done = true;
  }
  const int nwg = nwg16(B, heads, s);
  hipLaunchKernelGGL(attn16_fwd_kernel, dim3(nwg * heads), dim3(kThreadsFwd16), sm, s, (const __bf16*)qkv,
                     (long long)ldqkv, B, D, heads, table, (__bf16*)out, (long long)ldo, lse, nwg, opt(OPT_ATTN16_STAGGER_FWD));
  return check_launch("attn_fwd(14x14)");
}

int attn16_bwd(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const void* out, int64_t ldout, const float* lse,
               const float* delta, const float* table, int B, int D, int heads, float scale, void* dqkv, int64_t lddqkv,
               float* dtable, float* dq_bias, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
#define A16_ATTR(DT, FD) MEMHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn16_bwd_kernel<DT, FD>), \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds))
    A16_ATTR(true, true); A16_ATTR(true, false); A16_ATTR(false, true); A16_ATTR(false, false);
#undef A16_ATTR
    attr_done = true;
  }
  const int nwg = nwg16(B, heads, s), stagger = opt(OPT_ATTN16_STAGGER);
  const int grid = nwg * heads;
#define A16_LAUNCH(DT, FD)                                                                                          \
  hipLaunchKernelGGL((attn16_bwd_kernel<DT, FD>), dim3(grid), dim3(kThreads16), kLdsBwd16, s, (const __bf16*)qkv,     \
                     (long long)ldqkv, (const __bf16*)dout, (long long)ldo, (const __bf16*)out, (long long)ldout, lse, \
                     delta, table, (__bf16*)dqkv, (long long)lddqkv, dtable, dq_bias, B, D, heads, scale, nwg, stagger)
  // out != NULL: delta is computed inside the kernel (the `delta` workspace is not read)
  if (dtable) { if (out) A16_LAUNCH(true, true); else A16_LAUNCH(true, false); }
  else { if (out) A16_LAUNCH(false, true); else A16_LAUNCH(false, false); }
#undef A16_LAUNCH
  return check_launch("attn_bwd(14x14)");
}

}  // namespace memhip

