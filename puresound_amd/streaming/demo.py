"""Streaming TSE harness on the HIP path (mirror of DemoTseNet, egs/tse/demo/utils.py:47-128 of mcw519/PureSound),
for B concurrent streams.

Per 16-sample hop and stream: slide a 32-sample window, encode it to one frame, run one masker frame step, multiply,
decode the frame back to 32 samples, average the overlapping 16 samples with the previous output.  The B windows
are laid side by side as ONE signal [1, B*32] framed with hop = win = 32, so the encoder kernel emits the frame
axis = stream axis layout [1][C][ldB] the streaming masker works on, and the decoder kernel (hop = win: no
overlap) returns [1, B*32] = one 32-sample frame per stream.

hipGraphs: `streaming_inference` (one hop) replays one graph of encoder -> masker step -> mask -> decoder;
`streaming_inference_chunk` replays ONE graph per chunk (BASELINE configs[4]: 320 samples = 20 hops): the window
shifts, the 20 hop bodies, the averaging overlap-add and -- when the masker's segment counter crosses a boundary
inside the chunk -- the Mem-LSTM update at its hop are all inside the capture.  One graph per (hops, update position);
with 150-frame segments and 20-hop chunks that is three graphs.  The graphs are dropped when a parameter changes.
"""
from typing import Optional

import math

import torch
import torch.nn as nn

from .. import hip
from ..nnet.lobe.encoder import FreeEncDec
from .skim_inference import StreamingSkiM


def overlap_add(a: Optional[torch.Tensor], b: torch.Tensor, overlap_length: int) -> torch.Tensor:
    """egs/tse/demo/utils.py:121-128, on the last axis: the overlapped samples are averaged."""
    if a is None:
        return b
    squeeze = b.dim() == 1
    a2, b2 = (a.unsqueeze(0), b.unsqueeze(0)) if squeeze else (a, b)
    out = torch.cat([a2[:, :a2.shape[1] - overlap_length], hip.overlap_average(a2, b2, overlap_length)], dim=-1)
    return out[0] if squeeze else out


class DemoTseNet(nn.Module):
    """Encoder + streaming masker of the demo preset (utils.py:47-72)."""

    def __init__(self) -> None:
        super().__init__()
        self.encoder = FreeEncDec(win_length=32, hop_length=16, laten_length=128, output_active=True)
        self.masker = StreamingSkiM(input_size=128, hidden_size=256, output_size=128, n_blocks=4, seg_size=150,
                                    seg_overlap=False, causal=True, embed_dim=192, embed_norm=True,
                                    block_with_embed=[1, 1, 1, 1], embed_fusion="FiLM")
        self.training = False
        self.queue = None
        self.win_size = 32
        self.hop_size = 16
        self.ola_size = int(self.win_size - self.hop_size)
        self._graph = None

    def forward(self, noisy: torch.Tensor, embed: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def init_streams(self, streams: int = 1, use_graph: bool = True) -> None:
        """Reset the sliding windows and the masker state for `streams` concurrent streams."""
        self.queue = None
        self._graph = None
        self._chunk_graphs = {}
        self._tail = None
        self._sig = None
        self._use_graph = use_graph
        self.masker.init_status(streams=streams, use_graph=False)   # this harness captures the whole hop itself

    def _hop_body(self):
        m = self.masker
        b = m.streams
        wav = self.queue.reshape(1, b * self.win_size)
        feats, _ = hip.free_encode(wav, self.encoder.encoder.weight.detach(), self.win_size, True)   # [1, C, ldB]
        m._x_in = feats
        mask = m._frame_body()                                                                        # [1, C, ldB]
        out = hip.free_decode(feats, b, self.encoder.decoder.weight.detach(), self.win_size, mask, "linear", "none")
        return out.reshape(b, self.win_size)

    @torch.no_grad()
    def streaming_inference(self, chunk: torch.Tensor, embed: torch.Tensor) -> Optional[torch.Tensor]:
        """chunk [B,16] new samples per stream, embed [B,E] (or [E]) -> decoded frame [B,32]; None on the first hop
        (utils.py:78-98)."""
        hip.require_device(chunk, "DemoTseNet.streaming_inference")
        if embed.dim() == 1:
            embed = embed.unsqueeze(0)
        m = self.masker
        if not hasattr(m, "streams"):
            self.init_streams(chunk.shape[0])
        if self.queue is None:
            self.queue = torch.cat([torch.zeros_like(chunk), chunk], dim=-1).float().contiguous()
            return None
        self.queue[:, :self.hop_size] = self.queue[:, self.hop_size:].clone()
        self.queue[:, self.hop_size:] = chunk
        # embedding terms of the FiLM layers: refreshed in place when the embeddings change
        self._check_parameters()
        self._refresh_embedding(embed)
        if not self._use_graph:
            gen = self._hop_body()
        else:
            if self._graph is None:
                saved = [t.clone() for t in m._seg_h + m._seg_c]
                s = torch.cuda.Stream(chunk.device)
                s.wait_stream(torch.cuda.current_stream(chunk.device))
                with torch.cuda.stream(s):
                    self._hop_body()
                torch.cuda.current_stream(chunk.device).wait_stream(s)
                for t, v in zip(m._seg_h + m._seg_c, saved):
                    t.copy_(v)
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph):
                    self._gen = self._hop_body()
                for t, v in zip(m._seg_h + m._seg_c, saved):
                    t.copy_(v)
            self._graph.replay()
            gen = self._gen
        m.frames_counter += 1
        if m.frames_counter % m.seg_size == 0:
            m.update_mem_lstm()
            m.reset_seg_lstm_status()
            m.frames_counter = 0
        return gen.clone()

    def _check_parameters(self) -> None:
        """A captured graph replays the kernel plans' pointers: drop every graph when a parameter was updated."""
        sig = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if sig != self._sig:
            self._graph = None
            self._chunk_graphs = {}
            self._sig = sig
            self.masker._embed_key = None  # the per-frame conditioning terms were made with the old weights

    def _refresh_embedding(self, embed: torch.Tensor) -> None:
        m = self.masker
        key = (embed.data_ptr(), embed._version, tuple(embed.shape))
        if key != m._embed_key:
            if m._embed_static is None:
                m._embed_static = embed.detach().reshape(m.streams, -1).float().clone()
            else:
                m._embed_static.copy_(embed.reshape(m.streams, -1))
            m._embed_key = key
            for f in m.seg_input_fusion:
                if f is not None:
                    f.set_per_frame_condition(m._embed_static, m.embed_norm)

    def _chunk_body(self, hops: int, updates: tuple):
        """`hops` hops on the static buffers; the Mem-LSTM update + block-0 reset behind every hop listed in `updates`
        (skim_inference.py:205-218; a chunk longer than a segment wraps the frame counter more than once).  Only the masker step is recurrent: the windows of all hops are encoded in ONE
        launch up front (hop i of stream s = samples [16 i, 16 i + 32) of its previous half window followed by the
        chunk; one "utterance" per hop, so every hop's features are the [1, C, ldB] block the masker step reads), the
        masker steps write their masks side by side, and ONE decoder launch plus the averaging overlap-add of all hops
        follows.  Same arithmetic per frame as the hop loop.  Results are in _blocks / _tail / queue."""
        m, h, b = self.masker, self.hop_size, self.masker.streams
        if self.ola_size != h or self.win_size != 2 * h:  # (the batched kernels assume win = 2 hop, as in the reference harness)
            return self._chunk_body_by_hops(hops, updates)
        # window i of stream s = samples [16 i, 16 i + 32) of (second half of the previous window ‖ chunk): one launch
        hip.stream_windows(self.queue, self._chunk_in, self._wins, h)
        feats, _ = hip.free_encode(self._wins, self.encoder.encoder.weight.detach(), self.win_size, True)  # [hops,C,ldB]
        if self._masks is None or self._masks.shape != (hops, m.output_fc[1].out_channels, feats.shape[-1]):
            self._masks = torch.empty(hops, m.output_fc[1].out_channels, feats.shape[-1], dtype=torch.float32, device=feats.device)
            self._cores = torch.empty(hops, m.input_size, feats.shape[-1], dtype=torch.float32, device=feats.device)
        # block 0's input norm and the output layer do not touch the recurrent state either: one launch each
        x_ln = m.input_norm_all(feats) if m.block0_takes_input_norm() else None
        if x_ln is not None and self._wavefront and m.wavefront_ready():
            # the masker steps as a wavefront: cell (hop h, block i) needs (h, i-1) and (h-1, i) only, so the cells of an
            # anti-diagonal run side by side on parallel branches of the graph: hops + n_blocks - 1 dependent steps of
            # three kernels instead of hops * n_blocks (a Mem-LSTM update drains the pipeline: it touches every block)
            start = 0
            for u in list(updates) + [hops - 1]:
                self._wavefront_run(feats, x_ln, start, u + 1)
                if u in updates:
                    m.update_mem_lstm()
                    m.reset_seg_lstm_status()
                start = u + 1
                if start >= hops:
                    break
        else:
            for i in range(hops):
                m._x_in = feats[i:i + 1]
                m._frame_body(x_ln=None if x_ln is None else x_ln[i:i + 1], core_out=self._cores[i:i + 1])
                if i in updates:
                    m.update_mem_lstm()
                    m.reset_seg_lstm_status()
        m._output(self._cores, b, out=self._masks)
        frames = hip.free_decode(feats, b, self.encoder.decoder.weight.detach(), self.win_size, self._masks, "linear",
                                 "none")                                                  # [hops, B * win]
        # averaging overlap-add of all hops, the new tail and the new window queue: one launch
        hip.stream_overlap(frames, self._wins, self._tail, self._blocks, self.queue, h)

    _wavefront = True  # chunk body: (hop, block) cells as a wavefront (False: hop after hop)
    _batched_cells = True  # an anti-diagonal's cells as one launch per kernel (False: one graph branch per block, round 3)
    _wf_y = None

    def _wavefront_run(self, feats: torch.Tensor, x_ln: torch.Tensor, h0: int, h1: int) -> None:
        """Hops [h0, h1) of the chunk through all blocks as a wavefront: the cells (hop d - i, block i) of anti-diagonal d run
        side by side, block i on its own stream (the blocks' outputs are double buffered by hop parity)."""
        m = self.masker
        nb = m.n_blocks
        dev = feats.device
        if self._batched_cells and nb <= hip.MAX_CELLS:
            # the cells of an anti-diagonal differ in their pointers only: ONE launch per kernel and diagonal
            # (ps_*_cells_f32), so the chunk is a linear chain of 3 (hops + blocks - 1) launches on one stream -- no fork,
            # no join, no idle side streams
            if nb > 1 and (self._wf_y is None or len(self._wf_y) != nb - 1
                           or self._wf_y[0][0][0].shape != feats[0:1].shape):
                mk = lambda: torch.empty_like(feats[0:1])  # noqa: E731
                self._wf_y = [[(mk(), mk()) for _ in range(2)] for _ in range(nb - 1)]   # [block][hop parity] -> (y, y2)
            for d in range(h0, h1 + nb - 1):
                cells = []
                for i in range(nb):
                    h = d - i
                    if h0 <= h < h1:
                        cur = (feats[h:h + 1], x_ln[h:h + 1]) if i == 0 else self._wf_y[i - 1][h & 1]
                        cells.append((i, cur[0], cur[1], self._cores[h:h + 1], None if i == nb - 1 else self._wf_y[i][h & 1]))
                m._cells(cells)
            return
        main = torch.cuda.current_stream(dev)
        if getattr(self, "_wf_streams", None) is None or len(self._wf_streams) != nb:
            self._wf_streams = [torch.cuda.Stream(dev) for _ in range(nb)]
            self._wf_y = None
        if self._wf_y is None or self._wf_y[0][0][0].shape != feats[0:1].shape:
            mk = lambda: torch.empty_like(feats[0:1])  # noqa: E731
            self._wf_y = [[(mk(), mk()) for _ in range(2)] for _ in range(nb - 1)]   # [block][hop parity] -> (y, y2)
        # fork / join per anti-diagonal: the cells of step d go out on their blocks' streams and meet on the launch stream
        # again, so every dependency ((h, i-1) and (h-1, i) before (h, i); the reader of a buffer before its next writer)
        # is covered by the join of the step before.  (Cross edges between the side streams themselves -- an event wait
        # per dependency instead of a join per step -- made hipStreamEndCapture crash in ROCm 7.0's runtime.)
        for d in range(h0, h1 + nb - 1):
            cells = [(d - i, i) for i in range(nb) if h0 <= d - i < h1]
            if len(cells) == 1:     # the ramps of the wavefront: nothing to run beside it
                h, i = cells[0]
                cur = (feats[h:h + 1], x_ln[h:h + 1]) if i == 0 else self._wf_y[i - 1][h & 1]
                m._cell(i, cur[0], cur[1], self._cores[h:h + 1], None if i == nb - 1 else self._wf_y[i][h & 1])
                continue
            for h, i in cells:
                s = self._wf_streams[i]
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    cur = (feats[h:h + 1], x_ln[h:h + 1]) if i == 0 else self._wf_y[i - 1][h & 1]
                    m._cell(i, cur[0], cur[1], self._cores[h:h + 1], None if i == nb - 1 else self._wf_y[i][h & 1])
            for h, i in cells:
                main.wait_stream(self._wf_streams[i])

    _CHUNK_GRAPH_CAP = 24  # captured (hops, update pattern) variants kept; the oldest goes first

    def _updates_in_chunk(self, frames_counter: int, hops: int) -> tuple:
        """Hop indices of the chunk behind which the segment counter wraps (utils.py:100-118 / skim_inference.py:205-218)."""
        seg = self.masker.seg_size
        return tuple(i for i in range(hops) if (frames_counter + i + 1) % seg == 0)

    def _capture_chunk_graph(self, key: tuple, device) -> None:
        hops, updates = key
        m = self.masker
        state = m._seg_h + m._seg_c + [t for pair in m._mem_h + m._mem_c for t in pair] + [self.queue, self._tail]
        saved = [t.clone() for t in state]
        s = torch.cuda.Stream(device)
        s.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(s):          # warm-up outside the capture: plans, per-frame embedding terms
            self._chunk_body(hops, updates)
        torch.cuda.current_stream(device).wait_stream(s)
        for t, v in zip(state, saved):
            t.copy_(v)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._chunk_body(hops, updates)
        for t, v in zip(state, saved):
            t.copy_(v)
        while len(self._chunk_graphs) >= self._CHUNK_GRAPH_CAP:
            self._chunk_graphs.pop(next(iter(self._chunk_graphs)))
        self._chunk_graphs[key] = g

    def _chunk_body_by_hops(self, hops: int, updates: tuple):
        """The same, hop by hop (window shift, hop body, averaging overlap-add per hop)."""
        m, h = self.masker, self.hop_size
        for i in range(hops):
            self.queue[:, :h] = self.queue[:, h:].clone()
            self.queue[:, h:] = self._chunk_in[:, i * h:(i + 1) * h]
            frame = self._hop_body()
            self._blocks[:, i * h:(i + 1) * h] = hip.overlap_average(self._tail, frame, self.ola_size)[:, :h]
            self._tail.copy_(frame[:, h:])
            if i in updates:
                m.update_mem_lstm()
                m.reset_seg_lstm_status()

    @torch.no_grad()
    def streaming_inference_chunk(self, chunk: torch.Tensor, embed: torch.Tensor,
                                  pre_wav: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """chunk [B, n*16] -> the running output [B, L] (utils.py:100-118).  With graphs on, a chunk after the first is
        one graph replay; the result equals the hop-by-hop loop bit for bit."""
        hops = chunk.shape[-1] // self.hop_size
        m = self.masker
        if (not getattr(self, "_use_graph", False) or self.queue is None or pre_wav is None or hops == 0
                or pre_wav.shape[-1] < self.ola_size):
            for i in range(hops):
                cur = self.streaming_inference(chunk[:, i * self.hop_size:(i + 1) * self.hop_size], embed)
                if cur is not None:
                    pre_wav = overlap_add(pre_wav, cur, self.ola_size)
            return pre_wav
        hip.require_device(chunk, "DemoTseNet.streaming_inference_chunk")
        if embed.dim() == 1:
            embed = embed.unsqueeze(0)
        self._check_parameters()
        self._refresh_embedding(embed)
        b, h = m.streams, self.hop_size
        updates = self._updates_in_chunk(m.frames_counter, hops)
        if self._tail is None or getattr(self, "_chunk_in", None) is None or self._chunk_in.shape != (b, hops * h):
            dev = chunk.device
            self._chunk_in = torch.empty(b, hops * h, dtype=torch.float32, device=dev)
            self._blocks = torch.empty(b, hops * h, dtype=torch.float32, device=dev)
            self._tail = torch.empty(b, self.ola_size, dtype=torch.float32, device=dev)
            self._wins = torch.empty(hops, b * self.win_size, dtype=torch.float32, device=dev)
            self._masks = None
            self._chunk_graphs = {}
        self._chunk_in.copy_(chunk[:, :hops * h])
        self._tail.copy_(pre_wav[:, pre_wav.shape[-1] - self.ola_size:])
        key = (hops, updates)
        g = self._chunk_graphs.get(key)
        if g is None:
            # Capturing costs two runs of the chunk body (10 ms and more): the first graphed chunk of a given length pays it
            # for EVERY position pattern the segment counter can reach with chunks of that length, so that no later chunk
            # does (round 2 captured lazily: the 12-15 ms maxima of its latency lines were these captures).
            todo = [key]
            if not any(k[0] == hops for k in self._chunk_graphs):
                step = math.gcd(hops, m.seg_size)
                todo += [(hops, u) for u in sorted({self._updates_in_chunk(fc, hops)
                                                    for fc in range(m.frames_counter % step, m.seg_size, step)})
                         if (hops, u) != key]
            for k in todo[:self._CHUNK_GRAPH_CAP]:
                self._capture_chunk_graph(k, chunk.device)
            g = self._chunk_graphs[key]
        g.replay()
        m.frames_counter = (m.frames_counter + hops) % m.seg_size
        return torch.cat([pre_wav[:, :pre_wav.shape[-1] - self.ola_size], self._blocks, self._tail], dim=-1)
