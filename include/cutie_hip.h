/*
 * cutie_hip.h -- C ABI of libcutie_hip.so, the MI355X (gfx950) kernel library behind the
 * Cutie per-frame inference hot path (InferenceCore.step).
 *
 * The reference (hkchengrex/Cutie) has no FFI for this path: all arithmetic is delegated to
 * torch ops.  This header is therefore the boundary WE define under the
 * reference's Python surface (cutie.model.cutie.CUTIE / cutie.inference.inference_core.InferenceCore,
 * mirrored in cutie_amd/).  Each op below cites the reference torch-op sequence it replaces.
 *
 * Conventions
 *   - plain C, no torch types: raw device pointers + explicit sizes; the caller owns all memory
 *     (nothing is allocated inside); every entry point takes the hipStream_t to launch on (as void*).
 *   - return 0 on success, negative on error; cutie_hip_last_error() gives the message.
 *   - activations are NHWC ("pixel-major, channel-contiguous") bf16 unless stated; batch = objects.
 *   - a frame is executed as a *launch plan*: an array of cutie_op descriptors over a pre-allocated
 *     arena, replayed by cutie_exec() (one C call per stage; graph-capturable: no allocation, no sync).
 */
#ifndef CUTIE_HIP_H
#define CUTIE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CUTIE_OP_NI 24
#define CUTIE_OP_NF 6
#define CUTIE_OP_NP 16

typedef struct cutie_op {
    int32_t kind;               /* CUTIE_OP_* */
    int32_t flags;              /* op-specific bit flags */
    int32_t i[CUTIE_OP_NI];     /* integer parameters (per-op meaning below) */
    float f[CUTIE_OP_NF];       /* float parameters */
    uint64_t p[CUTIE_OP_NP];    /* device pointers */
} cutie_op;                     /* 256 bytes, mirrored by cutie_amd/ops.py:OP_DTYPE */

/* ---- conv flags ---- */
#define CUTIE_F_RELU_IN   1   /* relu applied to the input while loading (conv(F.relu(x))) */
#define CUTIE_F_OUT_F32   2   /* output stored as f32 (default bf16) */
#define CUTIE_F_RES_BCAST 4   /* residual has batch 1 and is broadcast over the B objects */
#define CUTIE_F_PLAIN     8   /* A/B switch: the general kernel where a specialised one exists (Cout == 1 on large maps) */
#define CUTIE_F_TILE_OFF  128 /* A/B switch: Cout == 1, Cin == 128 on large maps without the LDS-tiled kernel (conv_cout1_tile_kernel) */
#define CUTIE_F_PRIO      256 /* the launch belongs to the frame's critical path (the caller's stream): its waves raise their issue priority
                               * (s_setprio 1), so that look-ahead work sharing the compute units (window encoder, stacked read-outs) fills
                               * the gaps instead of taking every other issue slot.  CONV, and since ABI 4 UPSAMPLE2X_ADD, AREA_DOWN3, ECA_APPLY,
                               * GRU (four-channel form), UP4_SOFTMAX (four-pixel forms) and the chain forms of ATTN_Q2P / ATTN_SELF / ATTN_P2Q / QFFN
                               * (ABI 3 raised it unconditionally there); AFF_SCORE / AFF_SELECT / AFF_READOUT use flags&64 */
#define CUTIE_ACT_SHIFT   4   /* activation code in bits 4..6 */
#define CUTIE_ACT_NONE    0
#define CUTIE_ACT_RELU    1
#define CUTIE_ACT_SIGMOID 2
#define CUTIE_ACT_SQ1     3   /* x*x + 1   (shrinkage, big_modules.py:84) */

enum {
    /* CONV: y = act(conv2d(relu_in?(cat(x1,x2)), w) + bias + res)   -- implicit-GEMM on MFMA.
     * replaces nn.Conv2d / GConv2d (+ folded BatchNorm2d + ReLU + residual add) everywhere:
     * resnet.py:51-124, big_modules.py:36,81-87,207-235,257-306, group_modules.py:33-58,102-127,
     * channel_attn.py:25-29, modules.py:22-31,46-85, nn.Linear on pixel tokens (object_transformer.py,
     * transformer_layers.py, object_summarizer.py).
     * p0=x1 bf16 [B,H,W,ldx1]  p1=x2 bf16 [B,H,W,ldx2] (virtual channel concat, may be 0)
     * p2=w bf16 [CoutPad][Kpad] (k = (kh*KW+kw)*Cin + c, zero padded)  p3=bias f32[Cout] (may be 0)
     * p4=res bf16 [B|1,OH,OW,ldr] (may be 0)  p5=y bf16|f32 [B,OH,OW,ldy]
     * i: 0 B 1 H 2 W 3 C1 4 C2 5 ldx1 6 ldx2 7 OH 8 OW 9 Cout 10 ldy 11 KH 12 KW 13 stride 14 pad
     *    15 ldr 16 Kpad 17 tile id (table in conv_igemm.hip launch_conv / cutie_amd/ops.py TILES; 19 = Cout==1 kernel)
     *    18 real (un-padded) input channels: ignored by the kernel, used for flop accounting
     *    19 split-K factor (<=1: none; must divide Kpad/BK of the chosen tile).  grid.z slices the K tiles, every slice
     *       parks its fp32 partial tile in p6 and a second launch sums the slices in slice order (deterministic) and
     *       runs the epilogue.  p6 = f32 scratch, capacity i20*1024 floats >= splitk*M*roundup(Cout,8).
     *    side jobs (LDS-DMA tiles only): p7 = int64 [B,Cout] GAP accumulator the stored output is added to (x 2^24: every stored VALUE is
     *       converted to fixed point before anything is summed, so the accumulator is the same integer for every tile and arrival order);
     *       p8 = int64 buffer of i21 words cleared by this launch.
     *    p9 / i22, p10 / i23 (producer / consumer tiles 100.., else ignored; 0 = none): device ranges [p, p + bytes) that the
     *       producer waves read and discard once their last DMA piece is out -- the packed weights of the following conv(s)
     *       of the list, so that they are in every XCD's L2 when that launch starts.  Never changes a result.
     *    ABI 4 -- clips in lock step (several independent clips through one launch, batch = clips x objects; the reference runs one
     *       InferenceCore per video, eval_vos.py:97): with CUTIE_F_RES_BCAST and f0 > 0 the B objects come in groups of (int)f0 -- one clip
     *       each -- and group q adds ITS residual map [OH,OW,ldr] at p4 + q * (int)f1 rows (f1 >= OH*OW, a whole number; what
     *       MainToGroupDistributor('add') broadcasts per clip, group_modules.py:61-99).  f0 = 0: one map for all objects, as before. */
    CUTIE_OP_CONV = 1,
    /* MAXPOOL 3x3 s2 p1 (+relu if flags&1): resnet.py:131-134, big_modules.py:47-48,158-160
     * p0=x bf16 [B,H,W,C] p1=y bf16 [B,OH,OW,C]   i: 0 B 1 H 2 W 3 C 4 OH 5 OW */
    CUTIE_OP_MAXPOOL = 2,
    /* IMG_PREP: (image-mean)/std, zero-pad to /16, CHW f32 -> NHWC8 bf16; with masks also the
     * per-object [image, mask_k, others_k] stack of the mask encoder.
     * cutie.py:62,76,49-59; tensor_utils.py:7-22; big_modules.py:134-139
     * p0=image f32 [3,h0,w0] p1=masks f32 [K,H,W] (0 => K=1, image only) p2=y bf16 [K,H,W,8]
     * i: 0 h0 1 w0 2 H 3 W 4 pad_left 5 pad_top 6 K     f: 0..2 mean 3..5 std */
    CUTIE_OP_IMG_PREP = 3,
    /* UPSAMPLE2X_ADD: y = bilinear_x2(g, align_corners=False) + skip (broadcast over objects)
     * group_modules.py:19-23 + MainToGroupDistributor('add') modules.py:15-18
     * p0=g bf16 [B,h,w,C] p1=skip bf16 [1,2h,2w,C] p2=y bf16 [B,2h,2w,C]   i: 0 B 1 h 2 w 3 C
     * ABI 4: i4 > 0 = objects per clip (clips in lock step, see CONV): group q = b / i4 adds the skip map at p1 + q * i5 pixels (i5 >= 4hw) */
    CUTIE_OP_UPSAMPLE2X_ADD = 4,
    /* AREA_DOWN: r x r mean (F.interpolate mode='area', integer ratio) on NHWC bf16|f32 input
     * group_modules.py:26-30 (modules.py:59-60)
     * p0=x [B,H,W,ldx] (bf16, or f32 if flags&1) p1=y bf16 [B,H/r,W/r,ldy] (channels [0,C); channels
     * [C,Cz) are zero-filled)   i: 0 B 1 H 2 W 3 C 4 ldx 5 ldy 6 r 7 Cz */
    CUTIE_OP_AREA_DOWN = 5,
    /* MASK_DOWN: planar f32 masks [K,H,W] -> 16x area mean -> bf16 [K,h,w,8] = (mask, others, 0..)
     * plus plain f32 copy [K,h,w] of the down-sampled mask (object summarizer).
     * cutie.py:149-150,49-59; object_summarizer.py:62
     * p0=masks f32 [K,H,W] p1=y bf16 [K,h,w,8] p2=m16 f32 [K,h,w] (may be 0)  i: 0 K 1 H 2 W 3 r */
    CUTIE_OP_MASK_DOWN = 6,
    /* GAP: per-(b,c) mean over pixels (nn.AdaptiveAvgPool2d(1)), channel_attn.py:31-32
     * p0=x bf16 [B,HW,C] p1=y f32 [B,C] p2=part f32 [B,ceil(HW/64),C] (deterministic 2-stage sum)   i: 0 B 1 HW 2 C
     * flags&1: only the per-chunk partials are produced (ECA_APPLY finishes the reduction in its prologue) */
    CUTIE_OP_GAP = 7,
    /* ECA_APPLY: y = x * sigmoid(conv1d_k5(gap))[c] + r    channel_attn.py:33-37
     * p0=x bf16 [B,HW,C] p1=gap f32 [B,C] (written: the finished means) p2=w f32[5] p3=r bf16 [B,HW,C] p4=y bf16
     * p5=part f32 [B,ceil(HW/64),C] (GAP partials)   i: 0 B 1 HW 2 C
     * p6 (optional, C = 256) = packed weights bf16 [256] of a Cout = 1, 1x1 conv with fused input ReLU applied to y as stored, p7 = its bias f32 [1]
     *      (may be 0), p8 = its output f32 [B,HW]: the mask_pred head of a transformer block (object_transformer.py:151-164) without a launch */
    CUTIE_OP_ECA_APPLY = 8,
    /* GRU: h' = sig(f)*h*(1-sig(u)) + sig(u)*tanh(n), values=[f|u|n] f32   modules.py:35-43
     * p0=values f32 [B,HW,3C] p1=h f32 [B,HW,C] (in/out) p2=h_bf16 bf16 [B,HW,C] (out)  i: 0 n=B*HW 1 C
     * flags&1: one channel per thread (A/B switch; default: four, 16-byte accesses, when C % 4 == 0 and the buffers are aligned -- same bits) */
    CUTIE_OP_GRU = 9,
    /* SEG_AGG: prob=sigmoid(logit); aggregate -> (K+1) logits at 1/4 res   cutie.py:193-198,
     * tensor_utils.py:47-55      p0=logits f32 [K,h,w] p1=agg f32 [K+1,h,w]   i: 0 K 1 h*w */
    CUTIE_OP_SEG_AGG = 10,
    /* UP4_SOFTMAX: bilinear x4 (align_corners=False) of the K+1 logit planes, softmax over planes
     * cutie.py:199-200   p0=agg f32 [K+1,h,w] p1=prob f32 [K+1,4h,4w] p2=logits_up f32 (may be 0)
     * i: 0 K+1 1 h 2 w
     * flags&1: SEG_AGG fused -- p0 = raw logits f32 [K,h,w], the aggregation runs per tap inside the launch (K+1 <= 16) 
     * flags&2 (with flags&1): the one-pixel-per-thread form (A/B switch; default: four pixels per thread, P <= 8)
     * flags&4 (with flags&1, P <= 8, h, w % 4 == 0): the launch also produces MASK_DOWN(prob[1:], r = 16) for the next frame's pixel fusion:
     *      p3 = m16 f32 [K, H/16 * W/16], p4 = pair bf16 [K, H/16, W/16, i3] (channels 0, 1 written), i3 = channel pitch of pair (8 | 64);
     *      bit-identical to a MASK_DOWN launch on the stored probabilities
     * flags&16 (with flags&4): every lane aggregates its own six source pixels (A/B switch; default: the 6 x 6 source pixels under a 16 x 16
     *      output cell are aggregated once per wave and shared through LDS -- half the VALU instructions, same bits)
     * flags&8 (with flags&1): the kernels with a run-time object count (A/B switch; default: one instantiation per K = 1..7, whose loads are
     *      all in flight together -- same bits)
     * ABI 4: i4 > 1 = clips per launch (grid.y; the four-pixel forms only): clip c reads logits p0 + c*K*h*w and writes prob / logits_up
     *      + c*(K+1)*16hw, m16 + c*K*(hw/16), pair + c*K*(hw/16)*i3 -- per clip exactly the one-clip launch */
    CUTIE_OP_UP4_SOFTMAX = 11,
    /* MASK_MERGE: build the per-object mask planes of a frame with an input mask
     * inference_core.py:259-300.  plane t: src[t] >= 0 -> (idx==src[t]) [idx mode] / fmask[src[t]] [float
     * mode]; else (pred==0 ? 0 : (covered ? 0 : pred[t+1])).  Input mask is un-padded; planes are padded.
     * p0=idx i32 [h0,w0] | fmask f32 [n,h0,w0] p1=pred f32 [Kold+1,H,W] (may be 0) p2=src i32 [Knew]
     * p3=planes f32 [Knew,H,W]   i: 0 h0 1 w0 2 H 3 W 4 pad_left 5 pad_top 6 Knew 7 Kold 8 nfloat
     * flags&1: float mode */
    CUTIE_OP_MASK_MERGE = 12,
    /* AGG_SOFTMAX: prob = softmax(aggregate(planes))   inference_core.py:299-300
     * p0=planes f32 [K,HW] p1=prob f32 [K+1,HW]   i: 0 K 1 HW */
    CUTIE_OP_AGG_SOFTMAX = 13,
    /* LINEAR: small-M f32 linear for the object-query side (M = K*16 rows)
     * y = act(x @ W^T + b) + res     nn.Linear / MHA in/out projections, transformer_layers.py
     * p0=x f32 [M,ldx] p1=x_add f32 [M|16, Kd] (added to x before the product, may be 0)
     * p2=W bf16 [N,Kd] p3=bias f32 [N] p4=res f32 [M,N] (may be 0) p5=y f32 [M,ldy]
     * i: 0 M 1 N 2 Kd 3 ldx 4 ldy 5 add_rows (x_add row = m % add_rows)
     *    6 add_cols (> 0: x_add feeds only the output columns < add_cols, e.g. merged [q|k|v] projections where
     *      only q and k see the positional term; multiple of 16)
     * flags&1 relu   flags&2 fused nn.LayerNorm on x first (Kd == 256): p6=gamma f32[Kd] p7=beta f32[Kd]
     *      p8=ln_out f32 [M,Kd] (may be 0: the normalised rows, which the reference reuses as residual) f0=eps */
    CUTIE_OP_LINEAR = 14,
    /* LAYERNORM over the last dim (eps 1e-5)   p0=x f32 [M,C] p1=g p2=b f32[C] p3=y f32 [M,C]  i: 0 M 1 C */
    CUTIE_OP_LAYERNORM = 15,
    /* QUERY_INIT: obj_values = sums/(area+1e-4)   object_transformer.py:125-132
     * p0=obj_mem f32 [K,Q,C+1] p1=y f32 [K*Q,C]   i: 0 K*Q 1 C
     * flags&1 (C == 256, Q == 16): with the two linears that consume it (:137-138) -- p1=query p2=query_emb f32 [K*Q,256],
     *      p3/p4/p5 = W bf16 [256,256], bias, residual f32 [K*Q,256] of summary_to_query_init (+ query_init.weight), p6/p7/p8 of ..._emb
     *      side job: p9 (may be 0, 16-byte aligned) = a range of i2 x 16 bytes that is cleared (the accumulators of the chain forms below) */
    CUTIE_OP_QUERY_INIT = 16,
    /* AUX_MASK: foreground mask from mask_pred logits   object_transformer.py:179-205
     * p0=logits f32 [K,HW] p1=fg u8 [K,HW] p2=nfg i32 [K] (must be zeroed: MEMSET op before)
     * i: 0 K 1 HW */
    CUTIE_OP_AUX_MASK = 17,
    /* ATTN_Q2P: masked cross attention, 16 object queries <- HW pixels, 8 heads x 32
     * transformer_layers.py:45-98 (nn.MultiheadAttention core), object_transformer.py:56-61
     * p0=q f32 [K,Q,C] (projected) p1=kv bf16 [K,HW,ldkv] (k at +0, v at +voff) p2=fg u8 [K,HW]
     * p3=nfg i32 [K] p4=y f32 [K,Q,C]   i: 0 K 1 Q 2 HW 3 C 4 heads 5 ldkv 6 voff
     * flags&1: AUX_MASK fused -- p2=mask_pred logits f32 [K,HW] instead of fg, p3 unused (HW <= 24576)
     * flags&2 (with flags&1, C == 256): the q projection runs inside the launch -- p0=x f32 [K*Q, i7] unprojected rows, p3=ln_out f32
     *      [K*Q,256] (LayerNorm'd rows, may be 0), p5=Wq bf16 [C,256] p6=bias f32 [C] p7=query embedding f32 [K*Q,256] (may be 0)
     *      p8/p9=LayerNorm gamma/beta (0: no norm): q = (LN(x) + emb) Wq^T + b   (the LINEAR op it replaces: transformer_layers.py:86-93)
     * Chain form (flags&8, with flags 1|2; csrc/qchain.hip): the output projection runs inside the launch, per head, and is ADDED in
     *      fixed point to p13 = int64 [K*Q, 256] (value x 2^32; integer atomics: the sum is independent of the order of arrival; the
     *      caller clears it beforehand, e.g. QUERY_INIT's side job); p12 = Wo bf16 [256,256]; p4 is not written.  Wo's bias and the
     *      residual are added by the consumer through ITS flags&4:
     * flags&4 (chain form): the rows are a sum, x_eff = p0 + p11 (bias f32 [256], may be 0) + p10 / 2^32 (p10 = int64 [K*Q, 256])
     * i8 (chain form; 0 = 32): elements between the heads' k (and v) slices inside a pixel row -- 64 with i6 = 32 reads k | v interleaved per
     *      head (one 128-byte line per pixel and head)
     * flags&16 (chain form, instead of flags&4 and of the projection operands): p0 = q f32 [K*Q, 256], already projected and scaled by
     *      1/sqrt(32) (ATTN_P2Q flags&16 of the previous transformer block produces it); p3, p5..p11 unused
     * ABI 4 (chain form): i9 > 0 = objects per clip (clips in lock step): the foreground mask of object k is decided among the i9 objects
     *      of its own clip (logit planes (k / i9) * i9 ...), as object_transformer.py:179-205 aggregates over one video's objects */
    CUTIE_OP_ATTN_Q2P = 18,
    /* ATTN_SELF: 16x16 self attention per object  transformer_layers.py:12-41
     * p0=qk f32 [K,Q,ldqk] (q at +0, k at +C) p1=v f32 [K,Q,ldv] p2=y f32 [K,Q,C]
     * i: 0 K 1 Q 2 C 3 heads 4 ldqk (0: 2C) 5 ldv (0: C)
     * flags&2 (C == 256): the packed q|k|v in-projection runs inside the launch -- p0=x f32 [K*Q, i6], p3=ln_out (may be 0),
     *      p5=Wqkv bf16 [3C,256] p6=bias f32 [3C] p7=query embedding (q and k only) p8/p9=LayerNorm gamma/beta; p1 unused
     * flags&8 (+ optional flags&4): chain form as ATTN_Q2P (p10, p11 = accumulator input; p12, p13 = output projection into an accumulator;
     *      p2 is not written) */
    CUTIE_OP_ATTN_SELF = 19,
    /* ATTN_P2Q: pixels <- 16 queries cross attention  object_transformer.py:66-70
     * p0=q bf16 [K,HW,ldq] p1=kq f32 [K,Q,ldkv] p2=vq f32 [K,Q,ldkv] p3=y bf16 [K,HW,C]
     * i: 0 K 1 Q 2 HW 3 C 4 heads 5 ldq 6 ldkv (0: C)
     * flags&2 (C == 256): the packed k|v projection of the object queries runs inside the launch -- p1=x f32 [K*Q, i7],
     *      p5=Wkv bf16 [2C,256] p6=bias f32 [2C] p7=query embedding (k only); p2 unused
     * flags&4: chain form, accumulator input as ATTN_Q2P (p10, p11)
     * flags&16 (chain form): extra blocks also project the NEXT transformer block's ATTN_Q2P queries from the same rows x_eff:
     *      p15 = xn_out f32 [K*Q,256] = LN(x_eff; p8 gamma, p9 beta), p14 = q_out f32 [K*Q,256] = ((xn_out + p7) p12^T + p13) / sqrt(32)
     *      (p12 = Wq bf16 [256,256], p13 = bias) */
    CUTIE_OP_ATTN_P2Q = 20,
    /* SUMMARIZE: weights=sigmoid(logits)*[m x8 | (1-m) x8]; sums=einsum; area   object_summarizer.py:11-23
     * p0=feature bf16 [K,HW,C] p1=wlogits f32 [K,HW,Q] p2=m16 f32 [K,HW] p3=y f32 [K,Q,C+1]
     * p4=scratch f32 [K,ceil(HW/128),Q,C+1] (deterministic 2-stage sum)
     * flags&1: p0 is f32 with a row stride of i4 elements (0 = C); i5 = row stride of the logits (0 = Q) -- both read from one conv output
     * [feature | logits] (ABI 3)
     * i: 0 K 1 HW 2 C 3 Q 4 ldf 5 ldw */
    CUTIE_OP_SUMMARIZE = 21,
    /* ADD_PE: y = x + pe (broadcast over objects), bf16     object_summarizer.py:74-76
     * p0=x bf16 [B,HW,C] p1=pe bf16 [HW,C] p2=y   i: 0 B 1 HW*C */
    CUTIE_OP_ADD_PE = 22,
    /* KEY_PREP: split-bf16 MFMA operands of the anisotropic-L2 similarity  memory_utils.py:30-42
     * memory side (flags=0): A=[k^2|k] -> A_hi,A_lo bf16 [n,128], scale=shrinkage/sqrt(CK) f32 [n]
     * query side  (flags=1): B=[-e|2*k*e] -> B_hi,B_lo, c=sum(e*k^2)
     * p0=key f32 [n,64] p1=shr f32 [n] (mem) | sel f32 [n,64] (query) p2=hi p3=lo p4=scale|c   i: 0 n
     * flags&2 (query side): c summed by one lane per row in a 64-step loop (A/B switch; default: the row's lanes hand the running sum on --
     *      the same additions in the same order) */
    CUTIE_OP_KEY_PREP = 23,
    /* AFF_SCORE: S = scale_i*(A_i.B_j - c_j) tiles on MFMA (3-term split bf16, fp32-class accuracy)
     * mode 0: per-(16-token tile, query) maxima -> gmax f32 [HWp, Gld] (query-major, Gld = G rounded up to 64)
     * mode 1: append (S,token) with S >= tau_j to cand lists (f32,i32) [HW,cap], count i32 [HW*32] (counter of query j at [32*j]: one 128-B line each)
     * memory_utils.py:7-46 get_similarity + the candidate pre-filter of top-k (:58)
     * p0=A_hi p1=A_lo p2=scale (bank base pointers, rows = physical token slots) p3=B_hi p4=B_lo p5=c
     * p6=gmax | tau f32 [HW]  p7=cand_val p8=cand_idx p9=count
     * i: 0 HW 1 HWp 2 nranges 3.. (start,n) x3  9 G 10 cap 11 mode  12 query column sets (of 16) per wave: 1 | 2 (0 = 2) | 4 (aff_score4_kernel:
     *      256-query blocks, LDS-DMA staging; same bits for every choice)  13 tiles per block (0 = heuristic)  14 (diagnostic) KB of extra dynamic LDS  15 = 1: aff_score4_kernel (LDS-DMA staging) also for 2 sets per wave
     * flags&1 (mode 1): the gmax matrix of the mode-0 pass lies directly in front of tau in memory (p6 - HWp*Gld floats): every
     *      (16-token tile, 16-query set) whose maximum is below the set's thresholds is skipped (same result, ~1/6 of the MFMA work)
     * ABI 3 -- one read-out per bank version (memory_manager.py:112-208 reads once per frame; the bank changes on memory frames only,
     *      inference_core.py:238): i16 > 0 = query rows per frame; i1 (HWp) is then frames x i16 and the query operands of the frames are
     *      stacked (row j is a real query iff j % i16 < i0); c, gmax, tau, the candidate lists and counters are indexed by the stacked row.
     *      Per query the arithmetic is that of the one-frame launch (same bits).
     * ABI 4 -- clips in lock step (no counterpart in the reference: one InferenceCore, hence one bank, per video, eval_vos.py:97): flags&4 = the
     *      stacked frames read DIFFERENT banks, frame e the bank e % i17; p11 = u64 [i17][3] device table of the banks' (A_hi, A_lo, scale) bases
     *      (p0..p2 are ignored), the token ranges i3..i9 are those of every bank (banks on one schedule).  Needs 2 column sets per wave (i12),
     *      i16 % 128 == 0 (a block's 128 query rows lie inside one frame) and frames % i17 == 0; per query the bits of the one-frame launch
     *      on that frame's bank. */
    CUTIE_OP_AFF_SCORE = 24,
    /* AFF_SELECT: tau_j = top_k-th largest of gmax[:,j] (or -inf if G < top_k)
     * p0=gmax f32 [HWp,Gld] p1=tau f32 [HW]   i: 0 HW 1 HWp 2 G 3 top_k
     * optional side jobs (0 = none): p2=count i32 [HW*32]: count[32 q] = 0 for every query (pass 1's candidate counters);
     *      p3=life f32 [i4], p4=life f32 [i5]: += 1 (USAGE_TICK of two token ranges)
     * flags&1: range p3 is CLEARED instead (the usage side buffer of a look-ahead read-out, see USAGE_TICK)
     * flags&2: values per lane in the three sizes 16 | 32 | 64 only (A/B switch; default: ceil(G / 64) rounded up to a multiple of 4)
     * ABI 3: i6 > 1 = that many stacked frames of i1 query rows each (i0 real ones): gmax, tau and the counters are indexed by the stacked row */
    CUTIE_OP_AFF_SELECT = 25,
    /* AFF_READOUT: exact top-k of the candidates (ties -> lower slot), softmax, usage += w,
     * readout[o,j,:] = sum_i w_i V_o[i,:]    memory_utils.py:58-63,75; memory_manager.py:77-88
     * p0=cand_val p1=cand_idx p2=count p3=vptrs u64[K] (device array of per-object value-bank bases,
     * bf16 [slots,CV]) p4=usage f32 [slots] (may be 0) p5=y bf16 [K,HW,CV] p6=overflow i32[1]
     * i: 0 HW 1 cap 2 top_k 3 K 4 CV
     * ABI 3: i5 > 1 = that many stacked frames of i6 query rows each; frame f's read-out goes to y + f * K*HW*CV and its usage to
     *      p4 + f * i7 floats (per-frame side buffers: a look-ahead read-out is counted when -- and only when -- its frame is consumed)
     * ABI 4: i8 > 1 = that many banks, frame f gathers from bank f % i8: p3 = u64 [i8][K] (see AFF_SCORE flags&4)
     * ABI 4: flags&1 = p4 holds unsigned 64-bit FIXED-POINT counters (2^-40; i7 counts them) instead of f32: integer atomics commute, so the usage
     *      sums do not depend on the order in which the blocks arrive (the f32 form's last bits do); USAGE_TICK flags&1 adds them to the fp32 bank
     *      counters with one rounding.  The reference's usage is a dense column sum (memory_utils.py:58-63, kv_memory_store.py:151-162). */
    CUTIE_OP_AFF_READOUT = 26,
    /* MEMSET32: fill n 32-bit words with value i[1]   p0=dst   i: 0 n 1 value */
    CUTIE_OP_MEMSET32 = 27,
    /* COPY2D: rows x rowbytes strided device copy (bank append / compaction; replaces torch.cat,
     * kv_memory_store.py:7-16)   p0=src p1=dst   i: 0 rows 1 rowbytes(mult of 4) 2 src_stride 3 dst_stride */
    CUTIE_OP_COPY2D = 28,
    /* AXPY: y[i] += a * x[i] (f32)  streaming object-memory sum, memory_manager.py:264-269
     * p0=x p1=y  i: 0 n  f: 0 a */
    CUTIE_OP_AXPY = 29,
    /* USAGE_TICK: life[i] += 1 for i in [0,n)   kv_memory_store.py:161  p0=life  i: 0 n
     * optional (0 = none): p1=life2 f32 [i1]: += 1;  p2=use f32 [i2], p3=delta f32 [i2]: use += delta (the usage of a
     * look-ahead read-out, accumulated into a side buffer by AFF_READOUT and applied when the read-out is consumed)
     * ABI 4: flags&1 = p3 holds unsigned 64-bit fixed-point sums (2^-40, AFF_READOUT flags&1), converted with one rounding; flags&2 = and is
     *      cleared behind the addition (the side counters of a read-out that is always consumed) */
    CUTIE_OP_USAGE_TICK = 30,
    /* RANK_SELECT: order[r] = index of the r-th largest of use/life (ties -> lower index), r < k
     * torch.topk(usage, k) of memory_manager.py:339 and kv_memory_store.py:222
     * p0=use f32[n] p1=life f32[n] p2=order i32[k] p3=scratch i32[16*n] (partial ranks; required since ABI 3)
     * optional side jobs of the scattering launch: p4/p5, p6/p7 = src/dst of up to two row gathers dst[r,:] = src[order[r],:]
     * (rows of i2 / i3 32-bit words, multiples of 4: the prototype keys / selections of memory_manager.py:341-345);
     * p8 = u32 buffer of i4 words cleared (the column maxima of CONSOL_AFF)
     * i: 0 n 1 k 2 row words of gather 1 3 row words of gather 2 4 words to clear */
    CUTIE_OP_RANK_SELECT = 31,
    /* GATHER_ROWS: dst[r,:] = src[order[r],:]   p0=src p1=order i32 p2=dst  i: 0 k 1 rowbytes 2 src_stride 3 dst_stride */
    CUTIE_OP_GATHER_ROWS = 32,
    /* CONSOL_AFF (ABI 3): similarities of long-term consolidation, memory_manager.py:347-350 / memory_utils.py:30-42
     * S[p,i] = sim(cand_i, proto_p) for i < n, -inf for n <= i < ldS; colmax[p] = max_i S[p,i] as an order-preserving u32 key,
     * accumulated by atomicMax into a buffer the caller cleared (RANK_SELECT p8)
     * p0=ckey f32 [n,64] p1=cshr f32 [n] p2=pkey f32 [P,64] p3=psel f32 [P,64] p4=S f32 [P,ldS] p5=colmax u32 [P]
     * i: 0 n 1 P 2 ldS (multiple of 32, >= n) */
    CUTIE_OP_CONSOL_AFF = 33,
    /* CONSOL_READ (ABI 3): prototypes = softmax_i(S[p,:]) applied to the candidates' values of every object and to their shrinkage
     * memory_manager.py:347-356 (softmax over the candidates with max shift, memory_utils.py:68-71; aff @ values)
     * value rows of object o: candidates at vptrs[o] + (src + i) * C, prototypes written to vptrs[o] + (dst + p) * C (bf16);
     * out_shr[p] = sum_i softmax(S)[p,i] * cshr[i]
     * p0=S f32 [P,ldS] p1=colmax u32 [P] p2=vptrs u64 [K] p3=cshr f32 [n] p4=scratch f32 [nchunk * (K*P*C + 2*P)], nchunk = ceil(ldS/256)
     * p5=out_shr f32 [P] (or 0)
     * i: 0 n 1 P 2 C (multiple of 128) 3 K 4 ldS 5 src slot 6 dst slot */
    CUTIE_OP_CONSOL_READ = 34,
    /* CAST: f32 [n] -> bf16 [n] (flags=0) or bf16 -> f32 (flags=1)   p0=src p1=dst  i: 0 n */
    CUTIE_OP_CAST = 35,
    /* PROB_TO_ID: id = lut[argmax over the P planes] -- InferenceCore.output_prob_to_mask (inference_core.py:337-345) and
     * ResultSaver.process (results_utils.py:93-106: argmax + tmp-id -> object-id remap), fused; first maximum wins.
     * p0=prob f32 (P planes of H x W, plane stride i3 elements, row stride i4: the un-padded view that `step` returns
     * is addressed in place) p1=lut i32[P] p2=out [H,W] u8 (flags&3 == 0) | i32 (1) | i64 (2)
     * i: 0 P 1 H 2 W 3 plane stride 4 row stride */
    CUTIE_OP_PROB_TO_ID = 36,
    /* RESIZE: F.interpolate(x, size=(OH,OW)) -- the max_internal_size path of InferenceCore.step (inference_core.py:206-228,
     * 321-326): bilinear align_corners=False without antialias (flags&1 == 0) or nearest-exact (flags&1, index masks).
     * p0=src f32 (C planes of H x W, plane stride i5, row stride i6) p1=dst f32 [C,OH,OW]   i: 0 C 1 H 2 W 3 OH 4 OW 5 6 */
    CUTIE_OP_RESIZE = 37,
    /* FLIP_W: dst = f0 * flip_last_dim(src) + f1 * dst   -- torch.flip(x, dims=[-1]) of the flip_aug path and the averaging
     * of the two passes (inference_core.py:162-165,234-235,303-305).  src and dst must not overlap.
     * p0=src f32 [rows, W] (row stride i2) p1=dst f32 [rows, W] (row stride i3)   i: 0 rows 1 W 2 3   f: 0 alpha 1 beta */
    CUTIE_OP_FLIP_W = 38,
    /* AREA_DOWN3: three AREA_DOWNs in one launch (SensoryUpdater's area poolings of g8 / g4 / logits, modules.py:59-60).
     * segment q = 0..2: p[2q]=x p[2q+1]=y, i[8q..8q+7] = B H W C ldx ldy r Cz as in AREA_DOWN, flags bit q: f32 input;
     * flags&8: the bodies with a run-time r also for r = 2, 4 (A/B switch; default: r as a compile-time constant, all taps in flight -- same bits) */
    CUTIE_OP_AREA_DOWN3 = 39,
    /* QFFN: the FFN of a QueryTransformerBlock in one launch (transformer_layers.py:101-118: x + linear2(relu(linear1(norm(x))))),
     * split over FF / i2 slices of the hidden layer: grid (FF / i2, K).  Input rows as a sum (the self-attention out-projection):
     * x_eff = p0 (f32 [K*16,256], the residual) + p11 (bias f32 [256], may be 0) + p10 / 2^32 (p10 = int64 [K*16, 256], may be 0)
     * p1 = x_out f32 [K*16,256] (x_eff, written once: the residual of the FFN)  p2/p3 = LayerNorm gamma/beta
     * p4 = W1 bf16 [FF,256] p5 = b1 f32 [FF]  p6 = W2 bf16 [256,FF]
     * p7 = int64 [K*16, 256]: += relu(LN(x_eff) W1^T + b1) W2^T in fixed point (x 2^32), slice by slice (cleared by the caller).
     * linear2's bias and the residual x_out are added by the consumer (flags&4 of the attention ops).
     * i: 0 rows (K*16) 1 FF 2 hidden columns per block (64 | 128; 0: 64) */
    CUTIE_OP_QFFN = 40,
    /* STEM: IMG_PREP + the 7x7 / stride-2 / pad-3 conv with folded BN (Cin 3 + mask + others padded to 8, Cout 64) + 3x3 / stride-2 /
     * pad-1 max pool in one launch (resnet.py conv1, bn1, relu, maxpool; big_modules.py:30-33, 95-100): same rounding points as the
     * three ops (bf16 after the conv, max of bf16 values).
     * p0=image f32 [3,h0,w0] p1=masks f32 [K,H,W] (0: none, K = 1) p2=packed weights bf16 [64,Kpad] (k = (kh*7+kw)*8 + c)
     * p3=bias f32 [64] p4=y bf16 [K,H/4,W/4,64]   i: 0 h0 1 w0 2 H 3 W (multiples of 16) 4 pad_left 5 pad_top 6 K 7 Kpad
     * f: 0-2 mean 3-5 std   flags&1: ReLU (before or after the pool: the same)
     * ABI 4: i8 > 1 = frames per launch (<= 12; grid.y): frame f > 0 reads the image at p[4 + f] (p5 .. p15) and the masks at p1 + f * i9 floats and
     *      writes y + f * K*(H/4)*(W/4)*64 -- the frames of a look-ahead encoder window, or the clips of a lock-step group (one image + K masks each),
     *      in one launch that fills the chip (a 480p frame is 210 blocks: a partial round of its own); per frame exactly the one-frame launch */
    CUTIE_OP_STEM = 41,
    /* BANK_WRITE: the contiguous copies and fills of one memory insertion (memory_manager.py:210-296, kv_memory_store.py:55-149: the
     * reference torch.cat's every tensor of the bank; here a frame's keys / shrinkage / selection / per-object values go to their slot)
     * in ONE launch.  Up to 6 copies: p[2s] = src, p[2s+1] = dst, i[s] = 32-bit words (0: unused), s = 0..5; up to 2 fills:
     * p[12+t] = dst, i[6+t] = words, i[8+t] = the 32-bit pattern, t = 0..1.  Sources and destinations must not overlap. */
    CUTIE_OP_BANK_WRITE = 42,
    CUTIE_OP__COUNT
};

/* Execute n descriptors in order on `stream`.  Returns 0 or a negative error code. */
int cutie_exec(const cutie_op* ops, int n, void* stream);
/* Single-op entry (used by the kernel unit tests). */
int cutie_exec_one(const cutie_op* op, void* stream);
/* HIP-graph capture of a plan: returns an opaque handle (0 on failure); replay with cutie_graph_launch. */
void* cutie_graph_capture(const cutie_op* ops, int n, void* stream);
int cutie_graph_launch(void* graph, void* stream);
void cutie_graph_destroy(void* graph);
/* Timing helper: runs the plan `iters` times between two hipEvents on `stream`, returns mean ms (<0 on error).
 * bench.py uses it to time the dominant kernel on the launch stream (roofline.achieved). */
float cutie_time_ops(const cutie_op* ops, int n, int iters, void* stream);
const char* cutie_hip_last_error(void);
int cutie_hip_abi_version(void);
int cutie_op_struct_size(void);
/* bit 0: the library was built with -DCUTIE_DIAG (make DIAG=1): measured-and-lost kernel variants (ATTN_P2Q flags&32) are present.
 * The product library returns 0 and rejects those descriptors. */
int cutie_hip_build_flags(void);

#ifdef __cplusplus
}
#endif
#endif
