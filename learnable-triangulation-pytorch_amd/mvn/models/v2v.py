"""V2V-PoseNet 3D hourglass with the reference's surface (mvn/models/v2v.py of the reference):
``V2VModel(input_channels, output_channels)``, identical ``state_dict()`` keys.

Children are parameter containers; ``record()`` emits liblt_hip launches: channels-last (N,D,H,W,C)
implicit-GEMM conv3d on MFMA with BatchNorm3d + bias + ReLU + residual/skip adds folded into the epilogue,
2^3 stride-2 transposed convs as 8 parity phases of one launch, 2^3 max pools.
"""
import torch
import torch.nn as nn

import lt_engine as E
import lt_hip as H


def bn_tuple(bn):
    return E.BnParams(bn)


class Basic3DBlock(nn.Module):  # conv k^3 + BN + ReLU (reference :7-17)
    def __init__(self, in_planes, out_planes, kernel_size):
        super().__init__()
        self.block = nn.Sequential(nn.Conv3d(in_planes, out_planes, kernel_size, stride=1, padding=(kernel_size - 1) // 2),
                                   nn.BatchNorm3d(out_planes), nn.ReLU(True))

    def record(self, b, x):
        c = self.block[0]
        return b.conv(x, c.weight, c.bias, bn_tuple(self.block[1]), stride=1, pad=c.padding[0], relu=True)


class Res3DBlock(nn.Module):  # relu(BN(conv(relu(BN(conv x)))) + skip(x)) (reference :20-42)
    def __init__(self, in_planes, out_planes):
        super().__init__()
        self.res_branch = nn.Sequential(nn.Conv3d(in_planes, out_planes, 3, stride=1, padding=1), nn.BatchNorm3d(out_planes), nn.ReLU(True),
                                        nn.Conv3d(out_planes, out_planes, 3, stride=1, padding=1), nn.BatchNorm3d(out_planes))
        if in_planes == out_planes:
            self.skip_con = nn.Sequential()
        else:
            self.skip_con = nn.Sequential(nn.Conv3d(in_planes, out_planes, 1, stride=1, padding=0), nn.BatchNorm3d(out_planes))

    def record(self, b, x):
        r = self.res_branch
        skip = x
        if len(self.skip_con) and hasattr(b, "can_conv_skip") and b.can_conv_skip(tuple(x.shape[:-1]) + (r[0].weight.shape[0],), r[3].weight, x.shape,
                                                                                   self.skip_con[0].weight):
            # the 16 -> 32 block of the 64^3 level in bf16 plans: the skip convolution rides in the second convolution's launch (lt_conv_skip_fwd) --
            # its own launch, its 32-channel output and the read of that tensor disappear
            y = b.conv(x, r[0].weight, r[0].bias, bn_tuple(r[1]), stride=1, pad=1, relu=True)
            z = b.conv(y, r[3].weight, r[3].bias, bn_tuple(r[4]), stride=1, pad=1, relu=True,
                       skip=(x, self.skip_con[0].weight, self.skip_con[0].bias, bn_tuple(self.skip_con[1])))
            b.release(y)
            return z
        if len(self.skip_con):
            skip = b.conv(x, self.skip_con[0].weight, self.skip_con[0].bias, bn_tuple(self.skip_con[1]))
        y = b.conv(x, r[0].weight, r[0].bias, bn_tuple(r[1]), stride=1, pad=1, relu=True)
        z = b.conv(y, r[3].weight, r[3].bias, bn_tuple(r[4]), stride=1, pad=1, relu=True, residual=skip)
        b.release(y)
        if skip is not x:
            b.release(skip)
        return z


class Pool3DBlock(nn.Module):  # reference :45-51
    def __init__(self, pool_size):
        super().__init__()
        self.pool_size = pool_size

    def record(self, b, x):
        return b.maxpool(x, self.pool_size, self.pool_size, 0, nd=3)


class Upsample3DBlock(nn.Module):  # ConvTranspose3d k2 s2 + BN + ReLU (reference :54-66)
    def __init__(self, in_planes, out_planes, kernel_size, stride):
        super().__init__()
        assert kernel_size == 2 and stride == 2
        self.block = nn.Sequential(nn.ConvTranspose3d(in_planes, out_planes, kernel_size=2, stride=2, padding=0, output_padding=0),
                                   nn.BatchNorm3d(out_planes), nn.ReLU(True))

    def record(self, b, x, skip):
        """relu(BN(deconv x)) + skip: the decoder's skip add rides in the same epilogue (reference :121-136)."""
        c = self.block[0]
        return b.conv(x, c.weight, c.bias, bn_tuple(self.block[1]), stride=2, pad=0, transposed=True, relu_pre=True, residual=skip)


class EncoderDecorder(nn.Module):  # (sic) reference :69-138
    def __init__(self):
        super().__init__()
        chans = [32, 64, 128, 128, 128, 128]
        for lvl in range(1, 6):
            setattr(self, "encoder_pool%d" % lvl, Pool3DBlock(2))
            setattr(self, "encoder_res%d" % lvl, Res3DBlock(chans[lvl - 1], chans[lvl]))
        self.mid_res = Res3DBlock(128, 128)
        for lvl in range(5, 0, -1):
            setattr(self, "decoder_res%d" % lvl, Res3DBlock(chans[lvl], chans[lvl]))
            setattr(self, "decoder_upsample%d" % lvl, Upsample3DBlock(chans[lvl], chans[lvl - 1], 2, 2))
        for lvl in range(1, 6):
            setattr(self, "skip_res%d" % lvl, Res3DBlock(chans[lvl - 1], chans[lvl - 1]))

    def record(self, b, x):
        skips = []
        for lvl in range(1, 6):
            skips.append(getattr(self, "skip_res%d" % lvl).record(b, x))
            p = getattr(self, "encoder_pool%d" % lvl).record(b, x)
            b.release(x)
            x = getattr(self, "encoder_res%d" % lvl).record(b, p)
            b.release(p)
        y = self.mid_res.record(b, x); b.release(x); x = y
        for lvl in range(5, 0, -1):
            y = getattr(self, "decoder_res%d" % lvl).record(b, x); b.release(x)
            x = getattr(self, "decoder_upsample%d" % lvl).record(b, y, skips[lvl - 1])
            b.release(y); b.release(skips[lvl - 1])
        return x


class V2VModel(E.PlanCache):
    def __init__(self, input_channels, output_channels):
        super().__init__()
        self.front_layers = nn.Sequential(Basic3DBlock(input_channels, 16, 7), Res3DBlock(16, 32), Res3DBlock(32, 32), Res3DBlock(32, 32))
        self.encoder_decoder = EncoderDecorder()
        self.back_layers = nn.Sequential(Res3DBlock(32, 32), Basic3DBlock(32, 32, 1), Basic3DBlock(32, 32, 1))
        self.output_layer = nn.Conv3d(32, output_channels, kernel_size=1, stride=1, padding=0)
        for m in self.modules():  # reference :171-180
            if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)):
                nn.init.xavier_normal_(m.weight)
                nn.init.constant_(m.bias, 0)
        self.compute_dtype = torch.float32
        self._init_plan_cache()

    def record(self, b, x):
        """x: Act [N,V,V,V,Cin] -> logits Act [N,V,V,V,Cout], always fp32 (they feed the soft-argmax); the bf16 tail stores
        them planar (the Act's tensor is a channels-last view of (N,Cout,V,V,V) storage, see PlanBuilder.pwchain)."""
        for m in self.front_layers:
            y = m.record(b, x); b.release(x); x = y
        x = self.encoder_decoder.record(b, x)
        o = self.output_layer
        # the pointwise tail (back_layers[1:] + output_layer) runs as ONE pass over the volume when the plan is bf16
        tail = [m for m in list(self.back_layers)[1:]]
        chain = [(m.block[0].weight, m.block[0].bias, bn_tuple(m.block[1]), True) for m in tail] + [(o.weight, o.bias, None, False)]
        y = self.back_layers[0].record(b, x); b.release(x); x = y
        if all(isinstance(m, Basic3DBlock) and m.block[0].kernel_size == (1, 1, 1) for m in tail) and b.can_chain_pointwise(x, chain):
            y = b.pwchain(x, chain, planar=True)   # (N, J, V, V, V) storage: the soft-argmax reads whole volumes per joint
            b.release(x)
            return y
        for m in tail:
            y = m.record(b, x); b.release(x); x = y
        y = b.conv(x, o.weight, o.bias, None, out_f32=True)
        b.release(x)
        return y

    def forward(self, x):
        """x: (N,Cin,V,V,V) on the GPU -> (N,Cout,V,V,V) fp32 logits (reference :164-169)."""
        H.require_gpu(x, "volumes")
        if self.training:
            raise NotImplementedError("training runs through VolumetricTriangulationNet, whose step is recorded as a whole (lt_train.py); "
                                      "the stand-alone forward of this module is inference only: call .eval()")
        key = (tuple(x.shape), self.compute_dtype, x.device)

        def build():
            b = E.PlanBuilder(x.device, self.compute_dtype)
            inp = b.alloc((x.shape[0],) + tuple(x.shape[2:]) + (x.shape[1],))
            inp.pooled = False
            return {"plan": b.finish_after(self.record(b, inp)), "inp": inp}

        with torch.cuda.device(x.device):    # launches go to x's device whatever the caller's current device is
            P = self._plan_for(key, build)
            P["inp"].t.copy_(x.permute(0, 2, 3, 4, 1))
            P["plan"].run_eager(torch.cuda.current_stream(x.device).cuda_stream)
            return P["plan"].result.t.permute(0, 4, 1, 2, 3).clone(memory_format=torch.preserve_format)
