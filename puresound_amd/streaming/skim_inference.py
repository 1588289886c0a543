"""Streaming SkiM on the HIP path (mirror of puresound/streaming/skim_inference.py:10-252).

The reference streams ONE utterance: every state tensor has a constant batch axis of 1.  Here B concurrent
streams are the frame axis of the library's channel-major layout -- activations [1][C][ldB], LSTM states
[1][D*H][ldB] -- so one frame step of all streams is the same handful of kernels (FiLM conv, input projection,
ps_lstm_f32 with one step per stream, projection, LayerNorm + residual) whatever B is, and the whole step is
captured once in a hipGraph and replayed (no allocation, no host sync inside).  With B = 1 the tensors that
cross the API have exactly the reference's shapes.
"""
import time
from typing import List, Optional, Tuple

import torch

from .. import hip
from ..nnet._plans import lstm_path
from ..nnet.lobe.trivial import FiLM
from ..nnet.skim import SkiM


def _rows_to_frames(v: torch.Tensor) -> torch.Tensor:
    """[B, R] -> padded [1, R, ldB] (streams become frames)."""
    return hip.pad_rows(v.t().unsqueeze(0))


def _frames_to_rows(v: torch.Tensor, b: int) -> torch.Tensor:
    """padded [1, R, ldB] -> [B, R]."""
    return v[0, :, :b].t().contiguous()


class StreamingSkiM(SkiM):
    """Constructor as the reference (skim_inference.py:11-39)."""

    def __init__(self, input_size: int, hidden_size: int, output_size: int, n_blocks: int = 2, seg_size: int = 20,
                 seg_overlap: bool = False, causal: bool = True, embed_dim: int = 0, embed_norm: bool = False,
                 embed_fusion: Optional[str] = None, block_with_embed: Optional[List] = None, dropout: float = 0):
        super().__init__(input_size, hidden_size, output_size, n_blocks, seg_size, seg_overlap, causal, embed_dim,
                         embed_norm, embed_fusion, block_with_embed, dropout)
        self._graph = None

    # -- chunk API (stateless; the caller carries the states) -------------------------------------------
    @torch.no_grad()
    def step_chunk(self, x: torch.Tensor, seg_lstm_h_state=None, mem_lstm_h_hidden=None, seg_lstm_c_state=None,
                   mem_lstm_c_hidden=None, embed: Optional[torch.Tensor] = None):
        """The reference's chunk API (skim_inference.py:41-139) through torch.ops.puresound_amd.skim_chunk: the states the
        caller carries are flattened into one tensor list (absent ones as empty tensors) and come back the same way."""
        nb = self.n_blocks
        e = x.new_empty(0)
        flat = [e if seg_lstm_h_state is None else seg_lstm_h_state[i] for i in range(nb - 1)]
        flat += [e if seg_lstm_c_state is None else seg_lstm_c_state[i] for i in range(nb - 1)]
        for src in (mem_lstm_h_hidden, mem_lstm_c_hidden):
            for i in range(nb - 1):
                pair = None if src is None else src[i]
                flat += [e, e] if pair is None else [pair[0], pair[1]]
        from ..ops import call_args
        params, cfg = call_args(self, "skim_chunk")
        out = torch.ops.puresound_amd.skim_chunk(x, embed, flat, params, cfg)
        y, rest = out[0], list(out[1:])
        n1 = nb - 1
        seg_h, seg_c = rest[:n1], rest[n1:2 * n1]
        mem_h = [(rest[2 * n1 + 2 * i], rest[2 * n1 + 2 * i + 1]) for i in range(n1)]
        mem_c = [(rest[4 * n1 + 2 * i], rest[4 * n1 + 2 * i + 1]) for i in range(n1)]
        return y, seg_h, mem_h, seg_c, mem_c

    @torch.no_grad()
    def _step_chunk_impl(self, x: torch.Tensor, seg_lstm_h_state=None, mem_lstm_h_hidden=None, seg_lstm_c_state=None,
                         mem_lstm_c_hidden=None, embed: Optional[torch.Tensor] = None):
        """x [B,K,C] = one whole segment (B = 1 in the reference); states as the reference passes them: seg states
        lists of [D,B,H] for blocks 1.., Mem-LSTM hidden lists of ((h, c)) [D,B,H] (skim_inference.py:41-139).
        A frame-by-frame walk through the blocks equals one pass of every block's SegLSTM over the segment."""
        hip.require_device(x, "StreamingSkiM.step_chunk")
        b, k, c = x.shape
        nb, hid = self.n_blocks, self.hidden_size
        d = 1 if self.causal else 2
        to_state = lambda v: hip.pad_rows(v.permute(1, 0, 2).reshape(b, d * hid, 1).float())  # noqa: E731
        back = lambda v: v[..., 0].reshape(b, d, hid).permute(1, 0, 2).contiguous()  # noqa: E731
        if seg_lstm_h_state is not None and seg_lstm_c_state is not None:
            seg_h = [None] + [to_state(seg_lstm_h_state[i]) for i in range(nb - 1)]
            seg_c = [None] + [to_state(seg_lstm_c_state[i]) for i in range(nb - 1)]
        else:
            seg_h, seg_c = [None] * nb, [None] * nb
        if mem_lstm_h_hidden is None and mem_lstm_c_hidden is None:
            mem_h, mem_c = [None] * (nb - 1), [None] * (nb - 1)
        else:
            mem_h = [None if s is None else tuple(to_state(t) for t in s) for s in mem_lstm_h_hidden]
            mem_c = [None if s is None else tuple(to_state(t) for t in s) for s in mem_lstm_c_hidden]
        cur = hip.pad_rows(x.transpose(1, 2).float())          # [B, C, ldK]: every stream is an utterance
        for i in range(nb):
            cur = self._fuse(i, cur, k, embed)
            cur, (seg_h[i], seg_c[i]) = self.seg_lstm[i].forward_padded(cur, k, 1, k, seg_h[i], seg_c[i])
        out = hip.unpad_rows(self._output(cur, k), k)
        for i in range(nb - 1):
            seg_h[i], seg_c[i], mem_h[i], mem_c[i] = self.mem_lstm[i].forward_state(
                seg_h[i], seg_c[i], 1, mem_h[i], mem_c[i], want_states=True)
        return (out, [back(v) for v in seg_h[:-1]], [tuple(back(t) for t in s) for s in mem_h],
                [back(v) for v in seg_c[:-1]], [tuple(back(t) for t in s) for s in mem_c])

    # -- frame API (stateful) ------------------------------------------------------------------------------
    def init_status(self, streams: int = 1, device=None, use_graph: bool = True):
        """Initialise the streaming state (skim_inference.py:142-167) for `streams` concurrent streams."""
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("StreamingSkiM.init_status: ROCm device only (there is no CPU fallback)")
        d = 1 if self.causal else 2
        self.streams = streams
        self.frames_counter = 0
        ldb = hip.padded_frames(streams)
        z = lambda rows: torch.zeros(1, rows, ldb, dtype=torch.float32, device=dev)  # noqa: E731
        rows = d * self.hidden_size
        # causal: block i keeps [x'; h] in one row block so that its gates are one GEMM over K = C + H; the h state
        # is the lower part of that block
        self._xh = [z(self.input_size + rows) for _ in range(self.n_blocks)] if self.causal else None
        if self.causal:
            self._seg_h = [xh[:, self.input_size:, :] for xh in self._xh]
        else:
            self._seg_h = [z(rows) for _ in range(self.n_blocks)]
        self._seg_c = [z(rows) for _ in range(self.n_blocks)]
        self._h_new = [z(rows) for _ in range(self.n_blocks)]
        if self.causal:
            # each Mem-LSTM net keeps [x; h] in one row block too (x = the segment state it reads, h = its own LSTM state,
            # the lower part): its one-step update is the fused gates + cell kernel of the frame step, not an input GEMM and
            # a generic recurrence launch (64 us each at H = 256)
            self._mem_xh = [{k: z(2 * rows) for k in ("h", "c")} for _ in range(self.n_blocks - 1)]
            self._mem_h = [(xh["h"][:, rows:, :], z(rows)) for xh in self._mem_xh]
            self._mem_c = [(xh["c"][:, rows:, :], z(rows)) for xh in self._mem_xh]
            self._mem_hnew = z(rows)
        else:
            self._mem_xh = None
            self._mem_h = [(z(rows), z(rows)) for _ in range(self.n_blocks - 1)]
            self._mem_c = [(z(rows), z(rows)) for _ in range(self.n_blocks - 1)]
        self._x_in = z(self.input_size)
        self._embed_key = None
        self._embed_static = None
        for m in getattr(self, "seg_input_fusion", []):
            if m is not None:
                m._per_frame = None
        self._graph = None
        self._graph_sig = None
        self._use_graph = use_graph
        self._out = None
        print(f"{time.asctime(time.localtime(time.time()))}, Initialized streaming SkiM model")

    # reference-shaped views of the state (skim_inference.py:146-164), for B streams: [D, B, H]
    def _view(self, v: torch.Tensor) -> torch.Tensor:
        d = 1 if self.causal else 2
        return _frames_to_rows(v, self.streams).reshape(self.streams, d, self.hidden_size).permute(1, 0, 2).contiguous()

    def _assign(self, dst: torch.Tensor, value: torch.Tensor) -> None:
        """reference-shaped [D, B, H] (or [B, D*H]) state -> the frame-axis rows it lives in (in place: the captured
        frame graph keeps reading the same buffers)"""
        d = 1 if self.causal else 2
        v = torch.as_tensor(value, dtype=torch.float32, device=dst.device)
        if v.numel() != self.streams * d * self.hidden_size:
            raise ValueError(f"state of {tuple(v.shape)} does not hold {self.streams} streams x {d * self.hidden_size} values")
        if v.dim() == 3:
            v = v.permute(1, 0, 2)
        dst[0, :, :self.streams].copy_(v.reshape(self.streams, d * self.hidden_size).t())

    # the reference keeps these as plain assignable attributes (skim_inference.py:146-164): reading gives copies in its
    # layout, assigning a list of the same structure writes the values into the live state
    @property
    def seg_lstm_h_states(self):
        return [self._view(v) for v in self._seg_h]

    @seg_lstm_h_states.setter
    def seg_lstm_h_states(self, values):
        for dst, v in zip(self._seg_h, values):
            self._assign(dst, v)

    @property
    def seg_lstm_c_states(self):
        return [self._view(v) for v in self._seg_c]

    @seg_lstm_c_states.setter
    def seg_lstm_c_states(self, values):
        for dst, v in zip(self._seg_c, values):
            self._assign(dst, v)

    @property
    def mem_lstm_h_hidden(self):
        return [tuple(self._view(t) for t in s) for s in self._mem_h]

    @mem_lstm_h_hidden.setter
    def mem_lstm_h_hidden(self, values):
        for pair, vs in zip(self._mem_h, values):
            for dst, v in zip(pair, vs):
                self._assign(dst, v)

    @property
    def mem_lstm_c_hidden(self):
        return [tuple(self._view(t) for t in s) for s in self._mem_c]

    @mem_lstm_c_hidden.setter
    def mem_lstm_c_hidden(self, values):
        for pair, vs in zip(self._mem_c, values):
            for dst, v in zip(pair, vs):
                self._assign(dst, v)

    def reset_seg_lstm_status(self):
        self._seg_h[0].zero_()
        self._seg_c[0].zero_()

    def block0_takes_input_norm(self) -> bool:
        """block 0 starts with FiLM's own LayerNorm of the input frame (then `input_norm_all` may run it for many
        frames at once: it does not depend on the recurrent state)"""
        f = self.seg_input_fusion[0] if getattr(self, "seg_input_fusion", None) is not None else None
        return bool(self.causal and self._embed_static is not None and self.block_with_embed[0] and isinstance(f, FiLM)
                    and f.inp_norm and self.input_size % 2 == 0)

    def input_norm_all(self, frames: torch.Tensor) -> torch.Tensor:
        """[hops, C, ldB] encoder frames -> block 0's input norm of each, one launch"""
        f = self.seg_input_fusion[0]
        ln = f._plan_get(frames.device, f._build)["norm"]
        return hip.chan_layernorm(frames, self.streams, ln["gamma"], ln["beta"], ln["eps"])

    def wavefront_ready(self) -> bool:
        """Every block is the fused causal FiLM form (film kernel, gates + cell, projection + LayerNorm + the next block's
        input norm): then block i of hop h depends only on block i-1 of hop h and block i of hop h-1, and the demo harness
        runs the (hop, block) cells of a chunk as a wavefront on parallel graph branches."""
        if not self.causal or self._embed_static is None or self.input_size % 2:
            return False
        return all(self.block_with_embed[i] and isinstance(self.seg_input_fusion[i], FiLM)
                   and self.seg_input_fusion[i].inp_norm for i in range(self.n_blocks))

    def _cell(self, i: int, cur: torch.Tensor, cur_ln: torch.Tensor, core_out, y_bufs):
        """Block i of one hop on its own (wavefront_ready() form): cur / cur_ln = the block's input and its input norm;
        the output goes to core_out (last block) or y_bufs = (y, y2) static buffers.  Same three kernels as _frame_body."""
        b = self.streams
        c_in, hid = self.input_size, self.hidden_size
        rnn, proj, norm = self.seg_lstm[i].step_plan(cur_ln.device)
        xh = self._xh[i]
        x_rows, h_rows = xh[:, :c_in, :], xh[:, c_in:, :]
        self.seg_input_fusion[i].step_normed(cur_ln, b, x_rows)
        h_new = self._h_new[i]
        hip.lstm_gates_cell(xh, b, rnn["w_units"], rnn["bias_units"], self._seg_c[i], h_new, hid)
        norm2 = None
        if i + 1 < self.n_blocks:
            nxt = self.seg_input_fusion[i + 1]
            ln = nxt._plan_get(cur_ln.device, nxt._build)["norm"]
            norm2 = (ln["gamma"], ln["beta"], ln["eps"])
        last = core_out if i == self.n_blocks - 1 else y_bufs[0]
        hip.proj_layernorm(h_new, b, proj["wt"], proj["bias"], proj["M"], norm["gamma"], norm["beta"], norm["eps"], x_rows,
                           norm2, x_copy=h_rows, out=last, out2=None if norm2 is None else y_bufs[1])

    def _cells(self, cells: list) -> None:
        """Several _cell calls of DIFFERENT blocks -- [(i, cur, cur_ln, core_out, y_bufs), ...], the cells of one
        anti-diagonal of the wavefront -- as three launches in all (ps_*_cells_f32: blockIdx.y picks the cell) instead of
        three per cell on parallel streams.  Bit-identical to the per-cell launches (same kernels, same launch shapes)."""
        if len(cells) == 1:
            return self._cell(*cells[0])
        b = self.streams
        c_in, hid = self.input_size, self.hidden_size
        film, gates, proj_cells = [], [], []
        m_out = None
        for i, cur, cur_ln, core_out, y_bufs in cells:
            rnn, proj, norm = self.seg_lstm[i].step_plan(cur_ln.device)
            xh = self._xh[i]
            x_rows, h_rows = xh[:, :c_in, :], xh[:, c_in:, :]
            fusion = self.seg_input_fusion[i]
            film.append((cur_ln, fusion._plan_get(cur_ln.device, fusion._build)["wt_pairs"], fusion._per_frame_pairs, x_rows))
            gates.append((xh, rnn["w_units"], rnn["bias_units"], self._seg_c[i], self._h_new[i]))
            norm2 = None
            if i + 1 < self.n_blocks:
                nxt = self.seg_input_fusion[i + 1]
                ln = nxt._plan_get(cur_ln.device, nxt._build)["norm"]
                norm2 = (ln["gamma"], ln["beta"], ln["eps"])
            last = core_out if i == self.n_blocks - 1 else y_bufs[0]
            proj_cells.append(dict(x=self._h_new[i], wt=proj["wt"], bias=proj["bias"], gamma=norm["gamma"], beta=norm["beta"],
                                   eps=norm["eps"], res=x_rows, y=last, norm2=norm2,
                                   y2=None if norm2 is None else y_bufs[1], x_copy=h_rows))
            m_out = proj["M"]
        hip.film_conv_cells(film, b)
        hip.lstm_gates_cell_cells(gates, b, hid)
        hip.proj_layernorm_cells(proj_cells, b, m_out)

    def _frame_body(self, out: Optional[torch.Tensor] = None, x_ln: Optional[torch.Tensor] = None,
                    core_out: Optional[torch.Tensor] = None):
        """One frame of every stream through all blocks; reads _x_in / _embed_static, updates the seg states in
        place, returns padded [1, C_out, ldB] (written into `out` when given).  `x_ln`: block 0's input norm of _x_in,
        already computed (input_norm_all).  `core_out` [1, C, ldB]: the last block's output goes there and the output
        layer is left to the caller (one launch for many frames); returns None then."""
        b = self.streams
        c_in, hid = self.input_size, self.hidden_size
        cur, cur_ln = self._x_in, x_ln
        for i in range(self.n_blocks):
            fused = self._embed_static is not None and self.block_with_embed[i]
            if not self.causal:
                if fused:
                    cur = self.seg_input_fusion[i].forward_padded(cur, b, self._embed_static, self.embed_norm,
                                                                  per_frame=True)
                p = self.seg_lstm[i]._plan_get(cur.device, self.seg_lstm[i]._build)
                cur, _ = lstm_path(cur, b, *p, q=b, q_stride=1, steps=1, step_stride=0, h0=self._seg_h[i],
                                   c0=self._seg_c[i], state_out=(self._seg_h[i], self._seg_c[i]))
                continue
            rnn, proj, norm = self.seg_lstm[i].step_plan(cur.device)
            xh = self._xh[i]
            x_rows, h_rows = xh[:, :c_in, :], xh[:, c_in:, :]
            film = (fused and isinstance(self.seg_input_fusion[i], FiLM) and self.seg_input_fusion[i].inp_norm
                    and c_in % 2 == 0)      # the paired-row kernel needs an even channel count
            if film:
                if cur_ln is None:      # block 0 (or after an unfused block): the input norm as its own kernel
                    ln = self.seg_input_fusion[i]._plan_get(cur.device, self.seg_input_fusion[i]._build)["norm"]
                    cur_ln = hip.chan_layernorm(cur, b, ln["gamma"], ln["beta"], ln["eps"])
                self.seg_input_fusion[i].step_normed(cur_ln, b, x_rows)
            elif fused:
                self.seg_input_fusion[i].forward_padded(cur, b, self._embed_static, self.embed_norm, per_frame=True,
                                                        out=x_rows)
            else:
                x_rows.copy_(cur)
            # h' must not land in the rows the gates GEMM is still reading: it goes to a side buffer and the
            # projection kernel, which reads all of it anyway, hands it back to the [x'; h] block
            h_new = self._h_new[i]
            hip.lstm_gates_cell(xh, b, rnn["w_units"], rnn["bias_units"], self._seg_c[i], h_new, hid)
            nxt = self.seg_input_fusion[i + 1] if (i + 1 < self.n_blocks and self._embed_static is not None
                                                   and self.block_with_embed[i + 1]) else None
            norm2 = None
            if isinstance(nxt, FiLM) and nxt.inp_norm and c_in % 2 == 0:
                ln = nxt._plan_get(cur.device, nxt._build)["norm"]
                norm2 = (ln["gamma"], ln["beta"], ln["eps"])
            last = core_out if (core_out is not None and i == self.n_blocks - 1) else None
            cur, cur_ln = hip.proj_layernorm(h_new, b, proj["wt"], proj["bias"], proj["M"], norm["gamma"], norm["beta"],
                                             norm["eps"], x_rows, norm2, x_copy=h_rows, out=last)
        if core_out is not None:
            if cur.data_ptr() != core_out.data_ptr():
                core_out.copy_(cur)
            return None
        return self._output(cur, b, out)

    def state_tensors(self) -> List[torch.Tensor]:
        """The live streaming state in the library's layout (streams on the frame axis): seg h, seg c, Mem-LSTM (h, c) pairs
        of the h path, then of the c path -- what torch.ops.puresound_amd.skim_step updates in place."""
        return (list(self._seg_h) + list(self._seg_c) + [t for pair in self._mem_h for t in pair]
                + [t for pair in self._mem_c for t in pair])

    @torch.no_grad()
    def step_frame(self, x: torch.Tensor, embed: Optional[torch.Tensor]) -> torch.Tensor:
        """The reference's frame API (skim_inference.py:176-218) through torch.ops.puresound_amd.skim_step: x one frame per
        stream, embed [B,E] -> [B,C_out,1]; the state tensors are arguments of the operator and updated in place."""
        from ..ops import call_args
        params, cfg = call_args(self, "skim_step")
        fc = int(self.frames_counter)
        y = torch.ops.puresound_amd.skim_step(x, embed, self.state_tensors(), fc, params, cfg)
        self.frames_counter = (fc + 1) % self.seg_size
        return y

    @torch.no_grad()
    def _step_frame_impl(self, x: torch.Tensor, embed: Optional[torch.Tensor]) -> torch.Tensor:
        """x: one frame per stream, [B,C,1] / [B,1,C] / [B,C] (the reference passes [1,C,1] or [1,1,C]);
        embed [B,E] -> [B,C_out,1] (skim_inference.py:176-218)."""
        hip.require_device(x, "StreamingSkiM.step_frame")
        b = self.streams
        if x.numel() != b * self.input_size:
            raise RuntimeError(f"step_frame: expected {b} x {self.input_size} values, got {tuple(x.shape)}")
        self._x_in[0, :, :b].copy_(x.reshape(b, self.input_size).t())
        # a parameter update (load_state_dict, an in-place edit) rebuilds the kernel plans: the captured graph replays the
        # old pointers and the per-frame conditioning terms were made with the old weights -- both are redone
        sig = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if sig != self._graph_sig:
            self._graph = None
            self._graph_sig = sig
            self._embed_key = None
        if embed is not None:
            key = (embed.data_ptr(), embed._version, tuple(embed.shape))
            if key != self._embed_key:
                if self._embed_static is None:
                    self._embed_static = embed.detach().reshape(b, -1).float().clone()
                else:
                    self._embed_static.copy_(embed.reshape(b, -1))
                self._embed_key = key
                for m in getattr(self, "seg_input_fusion", []):
                    if m is not None:
                        m.set_per_frame_condition(self._embed_static, self.embed_norm)
        elif self._embed_static is not None:
            raise RuntimeError("step_frame: the stream was started with an embedding; keep passing it")
        if not self._use_graph:
            out = self._frame_body()
        else:
            if self._graph is None:
                # warm up once eagerly on a side stream (fills plan caches / per-frame embedding terms), then capture
                saved = [t.clone() for t in self._seg_h + self._seg_c]
                s = torch.cuda.Stream(x.device)
                s.wait_stream(torch.cuda.current_stream(x.device))
                with torch.cuda.stream(s):
                    self._frame_body()
                torch.cuda.current_stream(x.device).wait_stream(s)
                for t, v in zip(self._seg_h + self._seg_c, saved):
                    t.copy_(v)
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph):
                    self._out = self._frame_body()
                for t, v in zip(self._seg_h + self._seg_c, saved):
                    t.copy_(v)
            self._graph.replay()
            out = self._out
        res = out[0, :, :b].t().reshape(b, -1, 1).clone()
        self.frames_counter += 1
        if self.frames_counter % self.seg_size == 0:
            self.update_mem_lstm()
            self.reset_seg_lstm_status()
            self.frames_counter = 0
        return res

    @torch.no_grad()
    def update_mem_lstm(self):
        """skim_inference.py:220-252: block i's segment-end state -> MemLSTM i -> block i+1's initial state."""
        b = self.streams
        if self._mem_xh is not None and self.hidden_size == self.mem_lstm[0].h_net.input_size:
            # Blocks from the last to the first: Mem-LSTM i reads segment state i and writes segment state i+1, which Mem-LSTM
            # i+1 has read by then -- the reference's "compute all, then assign" without temporaries.  Per net: x rows <- the
            # segment state, gates + cell in one launch (cell state in place), projection + LayerNorm + residual straight into
            # the next block's state, h' handed back into the [x; h] block by the same launch.
            hid = self.hidden_size
            for i in range(self.n_blocks - 2, -1, -1):
                plans = self.mem_lstm[i].step_plans(self._seg_c[i].device)
                for key, src, dst, cell in (("h", self._seg_h[i], self._seg_h[i + 1], self._mem_h[i][1]),
                                            ("c", self._seg_c[i], self._seg_c[i + 1], self._mem_c[i][1])):
                    u, xh = plans[key], self._mem_xh[i][key]
                    xh[:, :hid, :].copy_(src)
                    hip.lstm_gates_cell(xh, b, u["w_units"], u["bias_units"], cell, self._mem_hnew, hid)
                    pr, nm = u["proj"], u["norm"]
                    hip.proj_layernorm(self._mem_hnew, b, pr["wt"], pr["bias"], pr["M"], nm["gamma"], nm["beta"], nm["eps"],
                                       xh[:, :hid, :], x_copy=xh[:, hid:, :], out=dst)
            return
        new = []
        for i in range(self.n_blocks - 1):
            new.append(self.mem_lstm[i].forward_state(self._seg_h[i], self._seg_c[i], b, self._mem_h[i],
                                                      self._mem_c[i], per_frame_sequences=True))
        for i, (h, c, hs, cs) in enumerate(new):
            self._seg_h[i + 1].copy_(h)
            self._seg_c[i + 1].copy_(c)
            for dst, src in zip(self._mem_h[i] + self._mem_c[i], hs + cs):
                dst.copy_(src)


# ---------------------------------------------------------------------------------------------------------------------
# the streaming API at the custom-op boundary: torch.ops.puresound_amd.skim_step / skim_chunk
# ---------------------------------------------------------------------------------------------------------------------
def _skim_step_hip(m: "StreamingSkiM", x, embed, state, frames_counter):
    b = x.numel() // m.input_size
    own = m.state_tensors() if getattr(m, "_seg_c", None) is not None and getattr(m, "streams", None) == b else None
    live = own is not None and len(own) == len(state) and all(p.data_ptr() == q.data_ptr() for p, q in zip(own, state))
    if not live:
        # a module rebuilt from a loaded trace, or a state set that is not this module's own: run on the module's buffers
        m.init_status(streams=b, device=x.device, use_graph=False)
        for dst, src in zip(m.state_tensors(), state):
            dst.copy_(src)
    m.frames_counter = int(frames_counter)
    y = m._step_frame_impl(x, embed)
    if not live:
        for dst, src in zip(state, m.state_tensors()):
            dst.copy_(src)
    return y


def _skim_step_meta(x, embed, state, frames_counter, params, cfg):
    import json
    ctor = json.loads(cfg).get("ctor") or {}
    b = x.numel() // int(ctor["input_size"])
    return x.new_empty((b, int(ctor["output_size"]), 1))


def _skim_chunk_hip(m: "StreamingSkiM", x, embed, states):
    n1 = m.n_blocks - 1
    opt = lambda t: None if t.numel() == 0 else t  # noqa: E731
    seg_h, seg_c = [opt(t) for t in states[:n1]], [opt(t) for t in states[n1:2 * n1]]
    pairs = lambda off: [None if states[off + 2 * i].numel() == 0 else (states[off + 2 * i], states[off + 2 * i + 1])  # noqa: E731
                         for i in range(n1)]
    mem_h, mem_c = pairs(2 * n1), pairs(4 * n1)
    none_if = lambda lst: None if all(v is None for v in lst) else lst  # noqa: E731
    y, sh, mh, sc, mc = m._step_chunk_impl(x, none_if(seg_h), none_if(mem_h), none_if(seg_c), none_if(mem_c), embed)
    return [y] + list(sh) + list(sc) + [t for pair in mh for t in pair] + [t for pair in mc for t in pair]


def _skim_chunk_meta(x, embed, states, params, cfg):
    import json
    ctor = json.loads(cfg).get("ctor") or {}
    b, k, _ = x.shape
    n1, hid = int(ctor["n_blocks"]) - 1, int(ctor["hidden_size"])
    d = 1 if ctor.get("causal", True) else 2
    st = lambda: x.new_empty((d, b, hid))  # noqa: E731
    return [x.new_empty((b, int(ctor["output_size"]), k))] + [st() for _ in range(6 * n1)]


def _register_ops():
    from ..ops import module_op
    module_op("skim_step", StreamingSkiM,
              "(Tensor x, Tensor? embed, Tensor(a!)[] state, int frames_counter, Tensor[] params, str cfg) -> Tensor",
              _skim_step_hip, _skim_step_meta)
    module_op("skim_chunk", StreamingSkiM,
              "(Tensor x, Tensor? embed, Tensor[] states, Tensor[] params, str cfg) -> Tensor[]",
              _skim_chunk_hip, _skim_chunk_meta)


_register_ops()
