"""Synthetic-data training harness: the body of the reference's train step
(trains/attention_scn.py:204-258, and the pure_scn / pure_attention siblings) with real flags instead
of module-level globals; defaults equal the reference's constants (:25-61).

One call of ``TrainStep.step()`` =
    encoder fwd -> decoder fwd -> CrossEntropy over packed tokens + alpha_c * doubly-stochastic term
    -> zero_grad -> backward (+ bucketed RCCL all-reduce when world > 1) -> clamp +-5 -> Adam.
Data loading and the ``.item()`` metric syncs of the reference are outside the step.
Tags are a synthetic input (the tagger encoder is a "next" row of SURVEY.md 8f)."""
import os

import torch
from torch import nn
from torch.nn.utils.rnn import pack_padded_sequence

from models.decoders.attention_scn import AttentionSCN
from models.decoders.pure_scn import PureSCN
from models.decoders.pure_attention import PureAttention
from models.encoders.caption import EncoderCaption
from models.encoders.tagger import EncoderTagger
from scnattn import functional as SF
from scnattn.dp import GradReducer, broadcast_parameters
from utils.optimizer import FusedClampAdam

DEFAULTS = dict(emb_dim=512, attention_dim=512, decoder_dim=512, factored_dim=512, semantic_dim=1000,
                dropout=0.5, batch_size=32, encoder_lr=1e-4, decoder_lr=4e-4, grad_clip=5.0, alpha_c=1.0,
                vocab_size=10000, max_len=50, image_size=256)


def synthetic_batch(batch_size, vocab_size, max_len, image_size, semantic_dim, device, seed, ragged=False):
    """Seeded synthetic inputs of SURVEY.md 8d: images ~ N(0,1); captions <start> + words + <end>
    padded with 0 to max_len+2 (word-map convention: pad=0, words 1.., unk, start=V-2, end=V-1)."""
    g = torch.Generator().manual_seed(seed)
    L = max_len + 2
    imgs = torch.randn(batch_size, 3, image_size, image_size, generator=g)
    tags = torch.rand(batch_size, semantic_dim, generator=g)
    caps = torch.zeros(batch_size, L, dtype=torch.long)
    lens = torch.full((batch_size,), L, dtype=torch.long)
    if ragged:
        lens = torch.randint(7, L + 1, (batch_size,), generator=g)
    for b in range(batch_size):
        n = int(lens[b])
        caps[b, 0] = vocab_size - 2
        caps[b, 1:n - 1] = torch.randint(1, vocab_size - 3, (n - 2,), generator=g)
        caps[b, n - 1] = vocab_size - 1
    return imgs.to(device), tags.to(device), caps.to(device), lens.unsqueeze(1).to(device)


def build_decoder(kind, cfg):
    if kind == "attention_scn":
        return AttentionSCN(cfg["attention_dim"], cfg["emb_dim"], cfg["decoder_dim"], cfg["factored_dim"],
                            cfg["semantic_dim"], cfg["vocab_size"], dropout=cfg["dropout"])
    if kind == "pure_scn":
        return PureSCN(cfg["emb_dim"], cfg["decoder_dim"], cfg["factored_dim"], cfg["semantic_dim"],
                       cfg["vocab_size"], dropout=cfg["dropout"])
    if kind == "pure_attention":
        return PureAttention(cfg["attention_dim"], cfg["emb_dim"], cfg["decoder_dim"], cfg["vocab_size"],
                             dropout=cfg["dropout"])
    raise ValueError("Error model type not found!")


_DIAG_SLEEP_CYCLES = int(os.environ.get("SCNATTN_DIAG_BWD_HEADSTART_CYCLES", "0"))


class TrainStep:
    def __init__(self, kind="attention_scn", fine_tune_encoder=True, device="cuda", seed=1234, encoder=True,
                 bucket_mb=32, graph_encoder=False, tagger=False, force_reduce=False, encoder_dtype="f32",
                 decoder_dtype="f32", tagger_overlap=True,
                 fused_loss=True, pooled_attention=True, **overrides):
        self.cfg = dict(DEFAULTS)
        self.cfg.update(overrides)
        self.kind = kind
        self.fused_loss = fused_loss
        self.pooled_attention = pooled_attention   # attention on the trunk's 8x8 map instead of its 14x14 pooling
        # "bf16": the ResNet trunk runs under bf16 autocast (MIOpen bf16 MFMA convolutions, bf16 feature maps
        # through the fused BatchNorm kernels, fp32 master weights/statistics); the decoder stays fp32.
        # This is BASELINE config 5's mixed-precision flavour, NOT the headline fp32 metric.
        self.encoder_bf16 = encoder_dtype == "bf16"
        # decoder "bf16": the operands the recurrence streams every step (recurrent weights, att1, the trunk map) are
        # read as bf16 copies made once per call (include/scnattn.h, option "decoder_bf16"); fp32 accumulate, fp32
        # softmax / LSTM state / master weights / gradients.  A process-wide option of the library.
        self.tagger_overlap = tagger_overlap
        self.decoder_bf16 = decoder_dtype in ("bf16", "bf16mfma")
        if torch.device(device).type == "cuda":      # "bf16mfma": also round the activation rows of the per-step products
            SF.set_option("decoder_bf16", {"bf16": 1, "bf16mfma": 2}.get(decoder_dtype, 0))   # and use the bf16 MFMA
        self.device = torch.device(device)
        torch.manual_seed(seed)  # same seed on every rank => identical initial weights
        self.decoder = build_decoder(kind, self.cfg).to(self.device)
        self.encoder = None
        self.encoder_optimizer = None
        if encoder:
            # Measured on MI355X (tools/miopen_probe.py, ResNet-152 fwd+bwd, B=32 fp32): channels-last with
            # MIOpen's find API (cudnn.benchmark) under FAST find mode = 40 ms/step after a one-off ~55 s of
            # kernel compilation; NCHW immediate mode = 50 ms; channels-last immediate mode = 320 ms.
            torch.backends.cudnn.benchmark = True
            self.encoder = EncoderCaption(channels_last=True).to(self.device)
            self.encoder.fine_tune(fine_tune_encoder)
            if fine_tune_encoder:
                self.encoder_optimizer = FusedClampAdam(
                    filter(lambda p: p.requires_grad, self.encoder.parameters()), lr=self.cfg["encoder_lr"],
                    grad_clip=self.cfg["grad_clip"])
        # the reference's real step also runs a frozen tagger ResNet-152 in train() mode to produce the tags
        # (trains/attention_scn.py:79-81, 194, 214); the benchmark feeds synthetic tags unless tagger=True
        self.tagger = None
        if tagger:
            self.tagger = EncoderTagger(semantic_size=self.cfg["semantic_dim"], channels_last=True).to(self.device)
            self.tagger.fine_tune(False)
            self.tagger.train()
        self.decoder_optimizer = FusedClampAdam(filter(lambda p: p.requires_grad, self.decoder.parameters()),
                                                lr=self.cfg["decoder_lr"], grad_clip=self.cfg["grad_clip"])
        self.criterion = nn.CrossEntropyLoss().to(self.device)
        self.reducers = [GradReducer(self.decoder_optimizer.flat, max(bucket_mb << 20, 4096))]
        broadcast_parameters(self.decoder_optimizer.flat)
        if self.encoder_optimizer is not None:
            self.reducers.append(GradReducer(self.encoder_optimizer.flat, max(bucket_mb << 20, 4096)))
            broadcast_parameters(self.encoder_optimizer.flat)
        if force_reduce:          # diagnostics: exercise hooks + collectives with a single rank
            for r in self.reducers:
                r.enabled = True
        self.encoder_events = None   # bench.py: a list -> every step appends (fwd start, fwd end, bwd start, bwd end) HIP events
        self.decoder.train()
        self.graphed = False         # make_graphed_callables patches the module's forward in place: no extra keywords then
        self.encoder_call = self.encoder
        if self.encoder is not None:
            self.encoder.train()   # reference quirk Q3: BN uses batch statistics even when frozen
            if graph_encoder and self.device.type == "cuda":
                # Optional: capture the encoder's forward and backward as two HIP graphs (static shapes).
                # Measured on MI355X / ROCm 7.2: graph replay of the ~2000-node encoder is SLOWER than eager
                # launches (51.5 vs 45.2 ms per step), so it is off by default.
                cfg = self.cfg
                sample = torch.randn(cfg["batch_size"], 3, cfg["image_size"], cfg["image_size"], device=self.device)
                self.encoder_call = torch.cuda.make_graphed_callables(self.encoder, (sample,), num_warmup_iters=3)
                self.graphed = True

    def loss_fn(self, scores, caps_sorted, decode_lengths, alphas, dl_dev=None):
        """trains/attention_scn.py:222-236.  On the GPU: one fused forward/backward pair on the unpacked
        scores (csrc/loss.hip); `fused_loss=False` keeps the reference's op sequence (tests compare the two)."""
        if self.fused_loss and scores.is_cuda:
            return SF.caption_loss(scores, caps_sorted, decode_lengths, alphas, self.cfg["alpha_c"], dl_dev)
        targets = caps_sorted[:, 1:]
        scores = pack_padded_sequence(scores, decode_lengths, batch_first=True).data
        targets = pack_padded_sequence(targets, decode_lengths, batch_first=True).data
        loss = self.criterion(scores, targets)
        if alphas is not None:
            loss = loss + self.cfg["alpha_c"] * ((1. - alphas.sum(dim=1)) ** 2).mean()
        return loss

    def step(self, imgs, tags, caps, caplens, encoder_out=None, prepool=None, drop_in=False, caplens_host=None):
        """`drop_in=True` is the reference's literal call sequence (trains/attention_scn.py:213-216):
        `encoder_out = encoder(imgs)` materialises the (B,14,14,2048) map, `decoder(encoder_out, ...)` receives only
        that tensor (and finds the trunk map EncoderCaption attached to it).  The default hands the trunk map over
        explicitly and skips the pooling kernel and its 51 MB write.  `caplens_host`: the caption lengths as a CPU tensor
        when the caller has them on the host anyway (a loader does): the decoder then needs no device synchronisation
        to learn its loop bounds (models/decoders/_common.py::sort_by_length); ignored for `drop_in`."""
        tag_event = None
        if self.tagger is not None and self.tagger_overlap and imgs.is_cuda and self.encoder is not None:
            # The frozen tagger's forward pass shares nothing with the caption encoder's: it runs on the side stream
            # (scnattn/conv.py: a stream probed to be concurrent with this one) beside it; the decoder is the first
            # consumer of the tags and waits for them by event.
            from scnattn import conv as _conv
            main = torch.cuda.current_stream(imgs.device)
            side = _conv._side(imgs.device)
            side.fork(main, imgs)
            with torch.cuda.stream(side.stream), torch.no_grad(), \
                    torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.encoder_bf16):
                tags = self.tagger(imgs).float()
            tag_event = torch.cuda.Event()
            tag_event.record(side.stream)
        ev = None
        if self.encoder_events is not None and self.encoder is not None and imgs.is_cuda:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.encoder_bf16):
            if self.encoder is not None:
                if self.pooled_attention and not self.graphed and not drop_in:
                    prepool = self.encoder(imgs, pooled=False)     # the decoder works on the 8x8 source map
                    encoder_out = None
                else:
                    encoder_out = self.encoder_call(imgs)
        if ev is not None:
            ev[1].record()
            feat = prepool if prepool is not None else encoder_out
            if feat.requires_grad:       # fires when the decoder's backward pass hands d(feature map) to the encoder's
                feat.register_hook(lambda g_, e=ev[2]: (e.record(), None)[1])
        if self.tagger is not None and tag_event is None:       # in line, after the caption encoder (no overlap)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.encoder_bf16):
                tags = self.tagger(imgs)
        if self.tagger is not None and tag_event is None:
            tags = tags.float()
        if tag_event is not None:
            torch.cuda.current_stream(imgs.device).wait_event(tag_event)
            tags.record_stream(torch.cuda.current_stream(imgs.device))
        if self.kind == "attention_scn":
            scores, caps_sorted, decode_lengths, alphas, sort_ind = self.decoder(
                encoder_out, tags, caps, caplens, prepool=prepool,
                pool_size=self.encoder.enc_image_size if self.encoder is not None else 14,
                caplens_host=None if drop_in else caplens_host)
        elif self.kind == "pure_scn":
            scores, caps_sorted, decode_lengths, sort_ind = self.decoder(
                encoder_out, tags, caps, caplens, prepool=prepool,
                pool_size=self.encoder.enc_image_size if self.encoder is not None else 14,
                caplens_host=None if drop_in else caplens_host)
            alphas = None
        else:
            scores, caps_sorted, decode_lengths, alphas, sort_ind = self.decoder(
                encoder_out, caps, caplens, prepool=prepool,
                pool_size=self.encoder.enc_image_size if self.encoder is not None else 14,
                caplens_host=None if drop_in else caplens_host)
        dl_dev = (caplens.reshape(-1)[sort_ind] - 1).to(torch.int32) if self.fused_loss else None
        loss = self.loss_fn(scores, caps_sorted, decode_lengths, alphas, dl_dev)
        self.decoder_optimizer.zero_grad()
        if self.encoder_optimizer is not None:
            self.encoder_optimizer.zero_grad()
        for r in self.reducers:
            r.reset()
        if _DIAG_SLEEP_CYCLES:      # diagnostics (tools/README.md): keep the GPU busy so that the host enqueues backward() ahead of it
            torch.cuda._sleep(_DIAG_SLEEP_CYCLES)
        loss.backward()
        if ev is not None:
            from scnattn import conv as _conv
            _conv.join_side_streams()       # the encoder's weight gradients on the side stream belong to its backward pass
            ev[3].record()
            self.encoder_events.append(ev)
        scale = 1.0
        for r in self.reducers:
            scale = r.finish()
        self.decoder_optimizer.step(scale)
        if self.encoder_optimizer is not None:
            self.encoder_optimizer.step(scale)
        return loss


def validate(batches, encoder, encoder_tagger, decoder, criterion, word_map, alpha_c=1.0):
    """The reference's validate() (trains/attention_scn.py:274-385): eval-mode forward under no_grad,
    loss / top-5 accuracy bookkeeping, references and arg-max hypotheses, corpus BLEU-4.
    `batches` yields (imgs, caps, caplens, allcaps) already on the device; `encoder_tagger` may be a
    callable returning tags or None when the batch tuple carries tags in place of images for it."""
    from utils.metric import AverageMeter, accuracy, corpus_bleu
    decoder.eval()
    encoder.eval()
    if hasattr(encoder_tagger, "eval"):
        encoder_tagger.eval()
    losses, top5accs = AverageMeter(), AverageMeter()
    references, hypotheses = [], []
    skip = {word_map['<start>'], word_map['<pad>']}
    with torch.no_grad():
        for imgs, caps, caplens, allcaps in batches:
            encoder_out = encoder(imgs)
            tags = encoder_tagger(imgs)
            scores, caps_sorted, decode_lengths, alphas, sort_ind = decoder(encoder_out, tags, caps, caplens)
            targets = caps_sorted[:, 1:]
            scores_copy = scores.clone()
            scores_p = pack_padded_sequence(scores, decode_lengths, batch_first=True).data
            targets_p = pack_padded_sequence(targets, decode_lengths, batch_first=True).data
            loss = criterion(scores_p, targets_p) + alpha_c * ((1. - alphas.sum(dim=1)) ** 2).mean()
            losses.update(loss.item(), sum(decode_lengths))
            top5accs.update(accuracy(scores_p, targets_p, 5), sum(decode_lengths))
            allcaps = allcaps[sort_ind]
            for j in range(allcaps.shape[0]):
                references.append([[w for w in c if w not in skip] for c in allcaps[j].tolist()])
            preds = torch.max(scores_copy, dim=2)[1].tolist()
            hypotheses.extend(p[:decode_lengths[j]] for j, p in enumerate(preds))
    assert len(references) == len(hypotheses)
    return corpus_bleu(references, hypotheses), losses.avg, top5accs.avg
