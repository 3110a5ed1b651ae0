"""Gate-block views of the SCN weights (reference: utils/tensor.py:1-42).

The four blocks are, in order, the input, forget, output and cell gates; 2-D weights are cut along
dim 1 unless ``front`` is set.  The HIP kernels never call these (they index the blocks in place);
they exist because callers and notebooks of the reference import them."""


def split_tensor1d(tensor, split):
    return [tensor[g * split:(g + 1) * split] if g < 3 else tensor[3 * split:] for g in range(4)]


def split_tensor2d(tensor, split, front=False):
    if front:
        return [tensor[g * split:(g + 1) * split, :] if g < 3 else tensor[3 * split:, :] for g in range(4)]
    return [tensor[:, g * split:(g + 1) * split] if g < 3 else tensor[:, 3 * split:] for g in range(4)]
