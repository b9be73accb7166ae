// The packed weight layout of the fused epoch kernels (smx_epoch.hip), shared with the optimiser kernel
// (smx_ppo.hip) that keeps it current.  Included inside an anonymous namespace.
//
// A matrix X [M, K] (a layer's weights, or their transpose for the backward kernel) is stored as
// [tile of 16 rows][32-wide K chunk][half][lane][4 floats], zero padded to whole tiles and an EVEN
// number of chunks: lane l = 16 kq + i of (tile t, chunk c, half h) holds X[16 t + i][32 c + 8 kq + 4 h + 0..3]
// -- the A operands of four consecutive v_mfma_f32_16x16x4_f32 steps, one contiguous KB per load
// instruction.  A net's copy is [W1 | W2 | W3 | W2^T | W3^T].
#pragma once

__host__ __device__ inline int pack_chunks(int K) { return (((K + 31) >> 5) + 1) & ~1; }
__host__ __device__ inline long pack_words(int M, int K) {       // 16-byte words of one packed block
    return (long)((M + 15) >> 4) * pack_chunks(K) * 128;
}
__host__ __device__ inline long pack_off(int D, int H1, int H2, int OUT, int blockno) {   // in 16-byte words
    long o = 0;
    if (blockno > 0) o += pack_words(H1, D);
    if (blockno > 1) o += pack_words(H2, H1);
    if (blockno > 2) o += pack_words(OUT, H2);
    if (blockno > 3) o += pack_words(H1, H2);
    if (blockno > 4) o += pack_words(H2, OUT);
    return o;
}
// float index of X[m][k] inside its packed block
__host__ __device__ inline long pack_pos(int K, int m, int k) {
    const int kk = k & 31;
    const long word = ((long)((m >> 4) * pack_chunks(K) + (k >> 5)) * 2 + ((kk >> 2) & 1)) * 64 + (kk >> 3) * 16 + (m & 15);
    return word * 4 + (kk & 3);
}
