"""Shared case tables for the emulator (CPU) and GPU kernel parity tests."""

# (x shape [major,h,w,minor], taps shape, up(x,y), down(x,y), pad(x0,x1,y0,y1))
UPFIRDN_SMALL = [
    ((6, 16, 16, 1), (4, 4), (1, 1), (1, 1), (2, 2, 2, 2)),      # D/Dpatch blur before 3x3 s2 conv
    ((6, 16, 16, 1), (4, 4), (1, 1), (1, 1), (1, 1, 1, 1)),      # blur before the 1x1 s2 skip / after G upsample
    ((3, 67, 70, 1), (4, 4), (1, 1), (1, 1), (2, 2, 2, 2)),      # non-multiple-of-tile plane
    ((5, 9, 9, 1), (3, 3), (1, 1), (1, 1), (0, 0, 0, 0)),        # E: [1,2,1] after reflection pad
    ((5, 8, 8, 1), (3, 3), (1, 1), (1, 1), (1, 0, 1, 0)),        # E skip: pad (1,0)
    ((4, 7, 7, 1), (1, 1), (1, 1), (1, 1), (0, 0, 0, 0)),        # E global branch blur [1]
    ((40, 4, 4, 1), (4, 4), (1, 1), (1, 1), (2, 2, 2, 2)),       # tiny planes, many of them
    ((3, 33, 33, 1), (4, 4), (1, 1), (1, 1), (1, 1, 1, 1)),
    ((3, 20, 12, 1), (2, 2), (1, 1), (1, 1), (1, 0, 1, 0)),
    ((3, 8, 8, 1), (4, 4), (2, 2), (1, 1), (2, 1, 2, 1)),        # Upsample (API parity)
    ((3, 16, 16, 1), (4, 4), (1, 1), (2, 2), (1, 1, 1, 1)),      # Downsample (API parity)
    ((2, 8, 8, 3), (4, 4), (1, 1), (1, 1), (2, 2, 2, 2)),        # minor > 1
    ((2, 9, 9, 1), (4, 4), (1, 1), (1, 1), (-1, -1, -1, -1)),    # negative pads crop
    ((2, 9, 7, 1), (5, 5), (1, 1), (1, 1), (2, 2, 2, 2)),        # > 4 taps (reference: uninitialised)
    ((2, 9, 7, 1), (3, 4), (2, 1), (1, 3), (2, 0, 1, 1)),        # anisotropic everything
    ((3, 130, 40, 1), (4, 4), (1, 1), (1, 1), (2, 2, 2, 2)),
    # x2 decimation / upsampling tile kernel: the skip-path pair (forward down 2, backward up 2)
    ((3, 64, 64, 1), (4, 4), (1, 1), (2, 2), (1, 1, 1, 1)), ((3, 32, 32, 1), (4, 4), (2, 2), (1, 1), (2, 2, 2, 2)),
    ((5, 37, 70, 1), (4, 4), (1, 1), (2, 2), (1, 1, 1, 1)), ((5, 19, 35, 1), (4, 4), (2, 2), (1, 1), (2, 3, 2, 3)),
    ((7, 9, 9, 1), (3, 3), (1, 1), (2, 2), (0, 0, 0, 0)), ((7, 4, 4, 1), (3, 3), (2, 2), (1, 1), (2, 2, 2, 2)),
    ((2, 257, 257, 1), (3, 3), (1, 1), (2, 2), (0, 0, 0, 0)),
    # 64-column tiles + a remainder of 1 ... 8 columns (blur_tail_kernel): 65, 129 + 2 wide outputs, 3 taps, row groups that
    # do not fill a unit of 32, more than one unit
    ((3, 20, 64, 1), (4, 4), (1, 1), (1, 1), (2, 2, 2, 2)), ((2, 17, 130, 1), (4, 4), (1, 1), (1, 1), (2, 2, 2, 2)),
    ((9, 40, 67, 1), (3, 3), (1, 1), (1, 1), (0, 0, 0, 0)), ((1, 300, 64, 1), (4, 4), (1, 1), (1, 1), (2, 2, 1, 1)),
]

# sae_upfirdn2d_epilogue_f32: (outer, channels, ih, iw, taps, up, pad(x0,x1,y0,y1)) -- the backward of the ResBlock's blur
# (4 taps, pad (2,2) forward -> pad (1,1) on the 2^k + 1 wide gradient), of the skip path's decimation (up 2), of the
# encoder's 3-tap blur; tiny planes (several strips per wave), odd sizes, one channel
K1_EPILOGUE = [
    (2, 3, 17, 17, 4, 1, (1, 1, 1, 1)), (1, 2, 67, 70, 4, 1, (1, 1, 1, 1)), (3, 5, 9, 9, 3, 1, (2, 0, 2, 0)),
    (4, 6, 5, 5, 4, 1, (1, 1, 1, 1)), (2, 4, 33, 33, 4, 1, (2, 2, 2, 2)), (1, 1, 40, 12, 2, 1, (0, 1, 0, 1)),
    (2, 3, 8, 8, 4, 2, (2, 1, 2, 1)), (1, 5, 19, 35, 4, 2, (2, 1, 2, 1)), (3, 4, 4, 4, 3, 2, (2, 2, 2, 2)),
    (2, 2, 32, 32, 4, 2, (2, 1, 2, 1)),
    (2, 3, 20, 66, 4, 1, (1, 1, 1, 1)), (1, 2, 70, 64, 4, 1, (2, 2, 2, 2)),      # remainder columns (blur_tail_kernel)
]

BIAS_ACT_SHAPES = [(2, 8, 16, 16), (3, 5, 7, 7), (4, 16), (2, 4, 33, 31), (2, 3, 32, 32), (5, 6, 20, 20),
                   (37, 6, 4, 4), (70, 40)]       # small planes, outer dimension cut into slices (column kernel)

# (n, c, h, w, m, k, stride, pad, weights stored [C,M,k,k])
CONV_SMALL = [
    (2, 8, 8, 8, 32, 3, 1, 1, False), (1, 5, 16, 16, 40, 3, 1, 1, False), (3, 10, 4, 4, 70, 3, 1, 1, False),
    (1, 9, 36, 33, 130, 3, 1, 0, False),
    (2, 8, 9, 9, 32, 3, 2, 0, False), (1, 5, 17, 17, 40, 3, 2, 0, False), (3, 10, 9, 9, 70, 3, 2, 0, False),
    (1, 6, 33, 33, 130, 3, 2, 0, False), (1, 4, 18, 18, 20, 3, 2, 0, False), (1, 4, 10, 12, 20, 3, 2, 1, False),
    (2, 40, 8, 8, 32, 1, 1, 0, False), (1, 70, 16, 16, 70, 1, 1, 0, False), (2, 33, 7, 7, 130, 1, 2, 0, False),
    (1, 20, 15, 15, 36, 1, 2, 0, False),
    (2, 8, 8, 8, 32, 3, 1, 1, True), (1, 6, 17, 17, 40, 3, 2, 0, True), (2, 8, 4, 4, 130, 3, 1, 0, False),
    (1, 3, 6, 6, 8, 3, 1, 1, False), (1, 3, 40, 40, 3, 1, 1, 0, False), (2, 6, 4, 4, 12, 3, 2, 0, False),
    # long K, few tiles: exercises the split-K (blockIdx.z) path of forward and stride-1 dgrad
    (2, 64, 8, 8, 40, 3, 1, 1, False), (1, 72, 9, 9, 40, 3, 2, 0, False), (1, 256, 4, 4, 40, 1, 1, 0, False),
    (2, 40, 6, 6, 70, 3, 1, 1, False),
    # stride-2 dgrad with 2^k + 1 wide half-resolution grids: main region + right/bottom strips
    (1, 4, 67, 67, 12, 3, 2, 0, False), (2, 4, 35, 67, 12, 3, 2, 0, False), (1, 4, 131, 35, 40, 3, 2, 0, False),
    # stride-2 dgrad / transposed conv producing > 64 channels: the class-split 128 x 128q transposed gather
    (1, 70, 17, 17, 12, 3, 2, 0, False), (2, 72, 35, 67, 8, 3, 2, 0, True), (1, 70, 10, 12, 20, 3, 2, 1, False),
    (1, 130, 9, 11, 6, 3, 2, 0, False),
    # stride-2 dgrad with a long channel loop and few tiles: the K split of conv_igemm_tr_kernel (slabs + fixed-order reduce),
    # also on a 7 x 7 -> 16 x 16 problem of the encoder's kind (pad 0, even input size) and with [C, M] weights
    (1, 12, 17, 17, 136, 3, 2, 0, False), (2, 8, 16, 16, 160, 3, 2, 0, False), (1, 20, 9, 9, 200, 3, 2, 0, True),
    # fp32 wgrad with vectorised staging (OW % 16|32 == 0): 16- and 32-column tiles, valid padding, M/C tails
    (2, 40, 16, 16, 40, 3, 1, 1, False), (1, 36, 32, 32, 70, 3, 1, 1, False), (1, 33, 18, 34, 40, 3, 1, 0, False),
    (1, 130, 8, 64, 36, 3, 1, 1, False),
    # narrow layers: wgrad MODE 1 (M, C <= 32) and MODE 2 (C * taps <= 32, RGB stems)
    (3, 20, 12, 12, 24, 3, 1, 1, False), (2, 3, 16, 16, 40, 3, 1, 1, False), (2, 3, 9, 9, 20, 3, 2, 0, False),
    (2, 3, 8, 8, 70, 1, 1, 0, False), (1, 30, 15, 15, 36, 1, 2, 0, False),
    # thin 1x1 layers on planes of >= 1024 pixels: the streaming weight gradient (conv1x1_thin_wgrad_kernel) -- FromRGB and ToRGB
    # shapes, a channel tail on the big side, [C, M] weights, two pixel slices
    (2, 3, 32, 32, 22, 1, 1, 0, False), (2, 21, 32, 32, 3, 1, 1, 0, False), (1, 2, 32, 40, 6, 1, 1, 0, True),
    (1, 4, 64, 64, 4, 1, 1, 0, False),
    # 128 x 128 tile with quad staging (rows a multiple of 4 wide, tiles 16 or 32 wide): valid padding with a partial
    # second tile column, two images of one 16 x 8 tile each with a channel tail, dgrad (pad 2) of a valid conv
    (1, 12, 20, 36, 70, 3, 1, 0, False), (2, 20, 16, 16, 130, 3, 1, 1, False), (1, 70, 10, 16, 9, 3, 1, 0, False),
    # stride-2 wgrad with 4-byte aligned quads: one partial 32-wide tile per row, the last quad of a row shifted left
    (2, 20, 21, 41, 40, 3, 2, 0, False), (2, 12, 21, 41, 70, 3, 2, 0, False),
]

# 3x3 stride-1 launches that take the 128x128 tile (the bf16x6 split-arithmetic kernel when
# sae_set_conv_math(1)): forward with M > 64, dgrad with C > 64, split-K tail, valid padding, [C,M] weights
CONV_BX = [
    (3, 10, 4, 4, 70, 3, 1, 1, False), (2, 64, 8, 8, 70, 3, 1, 1, False), (1, 9, 36, 33, 128, 3, 1, 0, False),
    (1, 70, 8, 8, 12, 3, 1, 1, False), (2, 72, 6, 6, 100, 3, 1, 1, True), (1, 5, 16, 40, 256, 3, 1, 1, False),
    # stride-2 dgrad / transposed conv producing > 64 channels: the 128 x 64q transposed-gather tile
    (1, 70, 17, 17, 12, 3, 2, 0, False), (2, 72, 35, 67, 8, 3, 2, 0, True), (1, 70, 10, 12, 20, 3, 2, 1, False),
    # narrow layers on the 64- and 32-row bf16x6 tiles
    (2, 24, 12, 12, 48, 3, 1, 1, False), (3, 20, 12, 12, 24, 3, 1, 1, False), (2, 64, 8, 8, 40, 3, 1, 1, False),
    # wgrad on octet tiles (OW % 8 == 0, OW >= 16): 16- and 32-column rows, valid padding, stride 2 (even/odd cells)
    (2, 40, 16, 16, 40, 3, 1, 1, False), (1, 36, 32, 32, 70, 3, 1, 1, False), (1, 33, 18, 34, 40, 3, 1, 0, False),
    (1, 40, 33, 33, 70, 3, 2, 0, False), (2, 33, 65, 65, 40, 3, 2, 0, False), (1, 36, 32, 32, 40, 3, 2, 1, True),
    (3, 130, 8, 24, 36, 3, 1, 1, False),
]

# larger shapes for the GPU (oracle still finishes in seconds): church-preset layer classes scaled down
CONV_GPU = CONV_SMALL + [
    (2, 128, 32, 32, 128, 3, 1, 1, False),     # D 3x3 s1
    (2, 64, 33, 33, 128, 3, 2, 0, False),      # D 3x3 s2 after blur
    (2, 64, 31, 31, 128, 1, 2, 0, False),      # D skip 1x1 s2
    (4, 32, 34, 34, 32, 3, 1, 0, False),       # E valid 3x3 after reflection pad (32-channel tile)
    (2, 64, 16, 16, 96, 3, 2, 0, True),        # G transposed 3x3 (weights [C,M]) via dgrad
    (8, 48, 4, 4, 96, 3, 1, 1, False),         # tails: many images per tile
    (2, 3, 64, 64, 32, 3, 1, 1, False),        # Dpatch stem
    (2, 128, 16, 16, 3, 1, 1, 0, False),       # ToRGB
    (3, 96, 4, 4, 48, 3, 1, 0, False),         # Dpatch final valid conv 4x4 -> 2x2
]

GEMM_CASES = [(16, 40, 300), (128, 70, 64), (5, 3, 1000), (70, 130, 33), (33, 200, 513),
              # aligned shapes: the 16-byte staging loads of both orientations (64 x 64 tile and the split-K 32 x 32 tile)
              (128, 256, 512), (64, 64, 96), (32, 128, 2048), (160, 96, 64)]
