"""2D backbone with the reference's surface (mvn/models/pose_resnet.py of the reference):
``get_pose_net(config, device)`` -> ``PoseResNet`` whose ``state_dict()`` keys/shapes equal the
reference's, so its checkpoints load with ``strict=True``.

The nn.Conv2d / nn.BatchNorm2d children are PARAMETER CONTAINERS ONLY: ``record()`` walks them once and
emits liblt_hip launches (implicit-GEMM convolutions on MFMA with BatchNorm, bias, residual and ReLU folded
into the epilogue, channels-last activations); ``forward`` replays the recorded plan.
"""
import torch
import torch.nn as nn

import lt_engine as E
import lt_hip as H

BN_MOMENTUM = 0.1

# depth -> (block kind, blocks per stage)            (reference :177-181)
resnet_spec = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3]),
               101: ("bottleneck", [3, 4, 23, 3]), 152: ("bottleneck", [3, 8, 36, 3])}


def _bn(c):
    return nn.BatchNorm2d(c, momentum=BN_MOMENTUM)


def bn_tuple(bn):
    return E.BnParams(bn)


class ResidualBlock(nn.Module):
    """Bottleneck (1x1, 3x3, 1x1; expansion 4) or BasicBlock (3x3, 3x3) parameter container.
    'simple' style strides the 3x3, 'caffe' style strides the first 1x1 (reference :57-137)."""

    def __init__(self, kind, inplanes, planes, stride=1, downsample=None, caffe=False):
        super().__init__()
        self.kind, self.stride = kind, stride
        if kind == "bottleneck":
            s1, s2 = (stride, 1) if caffe else (1, stride)
            shapes = [(inplanes, planes, 1, s1), (planes, planes, 3, s2), (planes, planes * 4, 1, 1)]
        else:
            shapes = [(inplanes, planes, 3, stride), (planes, planes, 3, 1)]
        self.strides = [s for _, _, _, s in shapes]
        for i, (ci, co, k, s) in enumerate(shapes, 1):
            setattr(self, "conv%d" % i, nn.Conv2d(ci, co, k, s, k // 2, bias=False))
            setattr(self, "bn%d" % i, _bn(co))
        self.downsample = downsample

    expansion = property(lambda self: 4 if self.kind == "bottleneck" else 1)

    def is_identity_bottleneck(self):
        return self.kind == "bottleneck" and self.downsample is None and all(s == 1 for s in self.strides)

    def record(self, b, x, t1=None, next_block=None):
        """Records the block over input ``x``.  ``t1``: this block's first activation relu(bn1(conv1(x))) when the PREVIOUS block's seam launch has already
        produced it (lt_expand_reduce_fwd); ``next_block``: fuse this block's expand with that block's reduce when the plan supports the shape -- the
        return value is then (y, t1 of the next block) instead of y (ResNet layer3's identity blocks: the 4 P-channel tensor is written once and not read
        back by the next block's first convolution)."""
        n = len(self.strides)
        if self.kind == "bottleneck" and self.downsample is None and t1 is None and next_block is None:
            # identity blocks of layer1 / layer2 in bf16 plans: the whole block in one launch, the two bottleneck-width tensors stay in LDS
            convs = [self.conv1.weight, self.conv2.weight, self.conv3.weight]
            if b.can_bottleneck(x, convs, self.strides):
                return b.bottleneck(x, convs, [bn_tuple(self.bn1), bn_tuple(self.bn2), bn_tuple(self.bn3)])
        if (self.kind == "bottleneck" and self.downsample is not None and t1 is None and next_block is None and hasattr(b, "can_bottleneck_ds")):
            # the first block of layer1 (downsample branch, stride 1) in bf16 plans: one launch, the residual branch computed from the tile of x in LDS
            convs = [self.conv1.weight, self.conv2.weight, self.conv3.weight]
            if b.can_bottleneck_ds(x, convs, self.strides, self.downsample[0].weight, self.downsample[0].stride[0]):
                return b.bottleneck_ds(x, convs, [bn_tuple(self.bn1), bn_tuple(self.bn2), bn_tuple(self.bn3)], self.downsample[0].weight,
                                       bn_tuple(self.downsample[1]))
        if t1 is not None or next_block is not None:
            assert self.is_identity_bottleneck()
            if t1 is None:
                t1 = b.conv(x, self.conv1.weight, None, bn_tuple(self.bn1), stride=1, pad=0, relu=True)
            t2 = b.conv(t1, self.conv2.weight, None, bn_tuple(self.bn2), stride=1, pad=1, relu=True)
            b.release(t1)
            if next_block is not None and b.can_expand_reduce(t2, x, self.conv3.weight, next_block.conv1.weight):
                y, t1n = b.expand_reduce(t2, x, self.conv3.weight, bn_tuple(self.bn3), next_block.conv1.weight, bn_tuple(next_block.bn1))
                b.release(t2)
                return y, t1n
            y = b.conv(t2, self.conv3.weight, None, bn_tuple(self.bn3), stride=1, pad=0, relu=True, residual=x)
            b.release(t2)
            return (y, None) if next_block is not None else y
        if self.kind == "bottleneck" and self.downsample is not None and hasattr(b, "can_conv_cat2"):
            # first block of layer2-4 in bf16 plans: the expand and the (strided) downsample branch are ONE pointwise convolution over [t2 | x] (lt_conv_cat2_fwd)
            ds, sds = self.downsample[0], self.downsample[0].stride[0]
            Ho, Wo = (x.shape[2] - 1) // sds + 1, (x.shape[3] - 1) // sds + 1
            if b.can_conv_cat2((x.shape[0], 1, Ho, Wo, self.conv3.weight.shape[1]), self.conv3.weight, x.shape, ds.weight, sds):
                t1 = b.conv(x, self.conv1.weight, None, bn_tuple(self.bn1), stride=self.conv1.stride[0], pad=0, relu=True)
                t2 = b.conv(t1, self.conv2.weight, None, bn_tuple(self.bn2), stride=self.conv2.stride[0], pad=1, relu=True)
                b.release(t1)
                y = b.conv_cat2(t2, self.conv3.weight, bn_tuple(self.bn3), x, ds.weight, bn_tuple(self.downsample[1]), sds)
                b.release(t2)
                return y
        res = x
        if self.downsample is not None:
            res = b.conv(x, self.downsample[0].weight, None, bn_tuple(self.downsample[1]), stride=self.downsample[0].stride[0], pad=0)
        y = x
        for i in range(1, n + 1):
            conv, bn = getattr(self, "conv%d" % i), getattr(self, "bn%d" % i)
            last = i == n
            z = b.conv(y, conv.weight, None, bn_tuple(bn), stride=conv.stride[0], pad=conv.padding[0], relu=True,
                       residual=res if last else None)
            if y is not x:
                b.release(y)
            y = z
        if res is not x:
            b.release(res)
        return y


class GlobalAveragePoolingHead(nn.Module):
    """Confidence head container (reference :140-174); see SURVEY.md section 8f row 3."""

    def __init__(self, in_channels, n_classes):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(in_channels, 512, 3, stride=1, padding=1), _bn(512), nn.MaxPool2d(2), nn.ReLU(inplace=True),
            nn.Conv2d(512, 256, 3, stride=1, padding=1), _bn(256), nn.MaxPool2d(2), nn.ReLU(inplace=True))
        self.head = nn.Sequential(nn.Linear(256, 512), nn.ReLU(inplace=True), nn.Linear(512, 256), nn.ReLU(inplace=True),
                                  nn.Linear(256, n_classes), nn.Sigmoid())

    def record(self, b, x):
        """x: layer4 output Act.  conv+BN (+ReLU, which commutes with the max pool) -> pool, twice; global
        average; three linears as 1x1 convolutions over an N-pixel map; sigmoid.  Returns Act [1,1,1,N,n]."""
        f = self.features
        y = b.conv(x, f[0].weight, f[0].bias, bn_tuple(f[1]), stride=1, pad=1, relu=True)
        p = b.maxpool(y, 2, 2, 0, nd=2); b.release(y)
        y = b.conv(p, f[4].weight, f[4].bias, bn_tuple(f[5]), stride=1, pad=1, relu=True); b.release(p)
        p = b.maxpool(y, 2, 2, 0, nd=2); b.release(y)
        g = b.global_avgpool(p); b.release(p)           # [1,1,1,N,256]
        h = self.head
        y = b.conv(g, h[0].weight[:, :, None, None], h[0].bias, None, relu=True); b.release(g)
        z = b.conv(y, h[2].weight[:, :, None, None], h[2].bias, None, relu=True); b.release(y)
        o = b.conv(z, h[4].weight[:, :, None, None], h[4].bias, None, sigmoid=True, out_f32=True); b.release(z)
        return o


class PoseResNet(E.PlanCache):
    def __init__(self, block_kind, layers, num_joints, num_input_channels=3, deconv_with_bias=False, num_deconv_layers=3,
                 num_deconv_filters=(256, 256, 256), num_deconv_kernels=(4, 4, 4), final_conv_kernel=1,
                 alg_confidences=False, vol_confidences=False, caffe=False):
        super().__init__()
        assert all(k == 4 for k in num_deconv_kernels), "only the 4x4 stride-2 deconvolution of the reference configs"
        self.num_joints, self.num_input_channels = num_joints, num_input_channels
        self.block_kind, self.caffe = block_kind, caffe
        exp = 4 if block_kind == "bottleneck" else 1
        self.conv1 = nn.Conv2d(num_input_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = _bn(64)
        inplanes = 64
        for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), layers), 1):
            stride = 1 if li == 1 else 2
            blocks = []
            for bi in range(nb):
                st = stride if bi == 0 else 1
                ds = None
                if bi == 0 and (st != 1 or inplanes != planes * exp):
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes * exp, kernel_size=1, stride=st, bias=False), _bn(planes * exp))
                blocks.append(ResidualBlock(block_kind, inplanes, planes, st, ds, caffe))
                inplanes = planes * exp
            setattr(self, "layer%d" % li, nn.Sequential(*blocks))
        if alg_confidences:
            self.alg_confidences = GlobalAveragePoolingHead(inplanes, num_joints)
        if vol_confidences:
            self.vol_confidences = GlobalAveragePoolingHead(inplanes, 32)
        dl = []
        for i in range(num_deconv_layers):
            dl += [nn.ConvTranspose2d(inplanes, num_deconv_filters[i], kernel_size=4, stride=2, padding=1, output_padding=0,
                                      bias=deconv_with_bias), _bn(num_deconv_filters[i]), nn.ReLU(inplace=True)]
            inplanes = num_deconv_filters[i]
        self.deconv_layers = nn.Sequential(*dl)
        self.final_layer = nn.Conv2d(inplanes, num_joints, kernel_size=final_conv_kernel, stride=1,
                                     padding=1 if final_conv_kernel == 3 else 0)
        self.compute_dtype = torch.float32
        self._init_plan_cache()

    # ---- plan recording -------------------------------------------------------------------
    def record(self, b, x, want_heatmaps=True, image_cell=None):
        """x: Act [N,1,H,W,Cpad] (zero-padded input channels).  Returns (heatmaps Act fp32 | None,
        features Act, alg_conf Act | None, vol_conf Act | None).  image_cell: see PlanBuilder.stem_pool (the fused stem reads
        the caller's fp32 images directly; x is then only a shape)."""
        if b.can_stem_pool(x, self.conv1.weight, 2, 3, (3, 2, 1)):   # bf16: conv1 + bn1 + relu + maxpool in one pass
            y = b.stem_pool(x, self.conv1.weight, bn_tuple(self.bn1), image_cell=image_cell)
        else:
            y = b.conv(x, self.conv1.weight, None, bn_tuple(self.bn1), stride=2, pad=3, relu=True)
            p = b.maxpool(y, 3, 2, 1, nd=2); b.release(y)
            y = p
        for li in range(1, 5):
            blocks = list(getattr(self, "layer%d" % li))
            t1 = None                                         # the next block's first activation, when the seam launch in front of it produced it
            for bi, blk in enumerate(blocks):
                nxt = blocks[bi + 1] if bi + 1 < len(blocks) else None
                # a run of identity bottleneck blocks whose seams lt_expand_reduce_fwd covers (ResNet layer3, bf16 plans): expand(i) + reduce(i + 1) in one launch
                chain = (nxt is not None and blk.is_identity_bottleneck() and nxt.is_identity_bottleneck() and hasattr(b, "can_expand_reduce") and
                         tuple(blk.conv3.weight.shape[:2]) == (1024, 256) and tuple(nxt.conv1.weight.shape[:2]) == (256, 1024) and b.dtype == torch.bfloat16 and
                         not getattr(b, "live_weights", False))
                if chain:
                    z, t1n = blk.record(b, y, t1=t1, next_block=nxt)
                elif t1 is not None:
                    z, t1n = blk.record(b, y, t1=t1), None
                else:
                    z, t1n = blk.record(b, y), None
                b.release(y)
                y, t1 = z, t1n
        alg = self.alg_confidences.record(b, y) if hasattr(self, "alg_confidences") else None
        vol = self.vol_confidences.record(b, y) if hasattr(self, "vol_confidences") else None
        for i in range(0, len(self.deconv_layers), 3):
            dc, bn = self.deconv_layers[i], self.deconv_layers[i + 1]
            z = b.conv(y, dc.weight, dc.bias, bn_tuple(bn), stride=2, pad=1, transposed=True, relu=True)
            b.release(y)
            y = z
        feats = y
        hm = None
        if want_heatmaps:
            fl = self.final_layer
            hm = b.conv(feats, fl.weight, fl.bias, None, stride=1, pad=fl.padding[0], out_f32=True)
        return hm, feats, alg, vol

    def heatmap_shape(self, h, w):
        def down(v, k, s, p):
            return (v + 2 * p - k) // s + 1
        for k, s, p in ((7, 2, 3), (3, 2, 1), (3, 2, 1), (3, 2, 1), (3, 2, 1)):
            h, w = down(h, k, s, p), down(w, k, s, p)
        return h * 8, w * 8

    # ---- stand-alone forward (reference :293-318) --------------------------------------------
    def forward(self, x):
        """x: (N,3,H,W) fp32 on the GPU -> (heatmaps (N,J,h,w), features (N,256,h,w), alg_confidences, vol_confidences)."""
        H.require_gpu(x, "images")
        if self.training:
            raise NotImplementedError("training runs through VolumetricTriangulationNet, whose step is recorded as a whole (lt_train.py); "
                                      "the stand-alone forward of this module is inference only: call .eval()")
        key = (tuple(x.shape), self.compute_dtype, x.device)
        N, Cc, Hh, W = x.shape

        def build():
            b = E.PlanBuilder(x.device, self.compute_dtype)
            inp = b.alloc((N, 1, Hh, W, E.min_cin_of(self.compute_dtype)))
            outs = self.record(b, inp, True)
            return {"plan": b.finish(), "inp": inp, "outs": outs}

        with torch.cuda.device(x.device):    # launches go to x's device whatever the caller's current device is
            P = self._plan_for(key, build)
            plan, inp, (hm, feats, alg, vol) = P["plan"], P["inp"], P["outs"]
            st = torch.cuda.current_stream(x.device).cuda_stream
            xin = x.float().contiguous()
            H.check(H.lib().lt_nchw_to_nhwc(plan_dtype_code(self.compute_dtype), xin.data_ptr(), inp.t.data_ptr(), N, Cc, Hh * W,
                                            inp.t.shape[-1], st), "lt_nchw_to_nhwc")
            plan.run_eager(st)
            to_nchw = lambda a: a.t[:, 0].permute(0, 3, 1, 2).to(torch.float32, copy=True)
            conf = lambda a: None if a is None else a.t.reshape(N, -1).clone()
            return to_nchw(hm), to_nchw(feats), conf(alg), conf(vol)


def plan_dtype_code(dt):
    return H.dtype_code(dt)


def get_pose_net(config, device="cuda:0"):
    """Same contract as the reference's get_pose_net (:321-377): builds the network for
    ``config.num_layers`` / ``config.style`` and optionally loads a pretrained checkpoint, keeping only
    shape-compatible tensors, stripping ``module.`` and partially copying the final layer."""
    kind, layers = resnet_spec[config.num_layers]
    if config.style == "caffe":
        kind = "bottleneck"   # the reference swaps in Bottleneck_CAFFE (expansion 4) at EVERY depth, 18 and 34 included (:322-324)
    model = PoseResNet(kind, layers, config.num_joints, num_input_channels=3, deconv_with_bias=False, num_deconv_layers=3,
                       num_deconv_filters=(256, 256, 256), num_deconv_kernels=(4, 4, 4), final_conv_kernel=1,
                       alg_confidences=config.alg_confidences, vol_confidences=config.vol_confidences,
                       caffe=(config.style == "caffe"))
    if config.init_weights:
        print("Loading pretrained weights from: {}".format(config.checkpoint))
        own = model.state_dict()
        loaded = torch.load(config.checkpoint, map_location="cpu")
        loaded = loaded.get("state_dict", loaded)
        picked = {}
        for k, v in loaded.items():
            name = k.replace("module.", "")
            if name in own and v.shape == own[name].shape:
                picked[name] = v
            elif name in ("final_layer.weight", "final_layer.bias"):
                print("Reiniting final layer:", k)
                o = torch.zeros_like(own[name])
                if name.endswith("weight"):
                    nn.init.xavier_uniform_(o)
                n = min(o.shape[0], v.shape[0])
                o[:n] = v[:n]
                picked[name] = o
        missing = {k.replace("module.", "") for k in loaded} - set(picked)
        if missing:
            print("Parameters [{}] were not inited".format(missing))
        model.load_state_dict(picked, strict=False)
        print("Successfully loaded pretrained weights for backbone")
    return model
