"""Quantized BART (encoder-decoder): where the quantizers sit.

Placement table (reference: quant_transformer/model/quant_bart.py):
  attention     q/k/v/out_proj -> QLinear (:120-123).  q*scaling, k, v quantized as [B,T,D] (seq axis 1)
                BEFORE the head split (:156-198); probs are the 3-D [B*h, T, S] tensor with seq axis 2 (:256);
                context [B,T,D] seq axis 1 (:272); out_proj output only if qoutput (:276)
  cross-attn    k/v come from the ENCODER states but are masked with the lengths the decoder layer passes,
                i.e. the decoder's (:167,172,472) -- kept
  probs mask    a length-B mask against B*h rows: observer.py:82's zip covers only the first B rows -- kept
                (ops.token_view(n_lengths=...))
  enc layer     self_attn(qoutput=False) -> GammaResidual -> LayerNorm(+q); fc1 -> act -> quantizer; fc2 ->
                GammaResidual -> LayerNorm(+q unless last layer of a qoutput=False stack) (:281-353)
  dec layer     self-attn block, cross-attn block (own GammaResidual + LayerNorm), ffn block (:369-492)
  embeddings    embed_tokens*embed_scale + learned positions (offset 2, own QEmbedding) -> layernorm_embedding (+q)
  model         shared / encoder.embed_tokens / decoder.embed_tokens are three separate QEmbedding copies
                (:553,698,901); encoder qoutput=True, decoder qoutput = model's (:904-907)
  lm            model(qoutput=True) -> lm_head (QLinear) + final_logits_bias (:1020-1023,1101)
Teacher-forced forward only (what calibration needs); generation stays with the FP HuggingFace model,
as in the reference driver (ptq_summ_quant.py:137-153).
"""
import torch
from torch import nn

from .losses import classification_loss, lm_loss, span_loss, with_loss
from ..quantization import QuantizedModule, Quantizer
from ..util_layernorm import (GammaResidual, QuantizedLayerNorm, activation_fake_quant, merge_heads_fake_quant,
                              qkv_heads_fake_quant, residual_layernorm, split_heads_fake_quant)


def shift_tokens_right(input_ids, pad_token_id, decoder_start_token_id):
    """quant_bart.py:24-37."""
    shifted = input_ids.new_zeros(input_ids.shape)
    shifted[:, 1:] = input_ids[:, :-1].clone()
    shifted[:, 0] = decoder_start_token_id
    shifted.masked_fill_(shifted == -100, pad_token_id)
    return shifted


def _causal_mask(bsz, tgt_len, dtype, device):
    """quant_bart.py:40-52: -inf above the diagonal."""
    mask = torch.full((tgt_len, tgt_len), float("-inf"), device=device)
    cond = torch.arange(tgt_len, device=device)
    mask.masked_fill_(cond < (cond + 1).view(tgt_len, 1), 0)
    return mask.to(dtype)[None, None, :, :].expand(bsz, 1, tgt_len, tgt_len)


def _expand_mask(mask, dtype, tgt_len=None):
    """quant_bart.py:55-66: [B,S] -> additive [B,1,T,S] with finfo.min on padding."""
    bsz, src_len = mask.shape
    tgt_len = tgt_len if tgt_len is not None else src_len
    inverted = 1.0 - mask[:, None, None, :].expand(bsz, 1, tgt_len, src_len).to(dtype)
    return inverted.masked_fill(inverted.bool(), torch.finfo(dtype).min)


def _plain_embedding(emb):
    """transformers >= 4.4x wraps BART's token embedding in a scaling nn.Embedding subclass; Quantizer() only
    recognises the exact nn.Embedding type (quantized_module.py:103-107), so rebuild a plain one on the same data."""
    if type(emb) is nn.Embedding:
        return emb
    plain = nn.Embedding(emb.num_embeddings, emb.embedding_dim, padding_idx=emb.padding_idx)
    plain.weight.data = emb.weight.data
    return plain


def _embed_scale(stack):
    """sqrt(d_model) if config.scale_embedding else 1 (stored on the stack in 4.18, on the embedding later)."""
    if hasattr(stack, "embed_scale"):
        return stack.embed_scale
    return getattr(stack.embed_tokens, "embed_scale", 1.0)


class QuantizedBartLearnedPositionalEmbedding(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.offset = 2
        self.qoutput = qoutput
        n, d = org_module.weight.shape
        plain = nn.Embedding(n, d)
        plain.weight.data = org_module.weight.data.clone()
        self.position_embeddings = Quantizer(plain, w_qconfig)

    def forward(self, input_ids_shape, past_key_values_length=0):
        seq_len = input_ids_shape[1]
        positions = torch.arange(past_key_values_length, past_key_values_length + seq_len, dtype=torch.long,
                                 device=self.position_embeddings.weight.device)
        return self.position_embeddings(positions + self.offset)


class QuantizedBartAttention(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.qoutput = qoutput
        self.embed_dim, self.num_heads, self.head_dim = org_module.embed_dim, org_module.num_heads, org_module.head_dim
        self.dropout = org_module.dropout
        self.scaling = self.head_dim ** -0.5
        self.is_decoder = org_module.is_decoder
        self.k_proj = Quantizer(org_module.k_proj, w_qconfig)
        self.v_proj = Quantizer(org_module.v_proj, w_qconfig)
        self.q_proj = Quantizer(org_module.q_proj, w_qconfig)
        self.out_proj = Quantizer(org_module.out_proj, w_qconfig)
        self.query_post_act_fake_quantize = Quantizer(None, a_qconfig)
        self.key_post_act_fake_quantize = Quantizer(None, a_qconfig)
        self.value_post_act_fake_quantize = Quantizer(None, a_qconfig)
        self.attention_probs_post_act_fake_quantize = Quantizer(None, a_qconfig)
        self.context_post_act_fake_quantize = Quantizer(None, a_qconfig)
        if qoutput:
            self.out_proj_post_act_fake_quantize = Quantizer(None, a_qconfig)

    def forward(self, hidden_states, key_value_states=None, attention_mask=None, observation_mask=None):
        bsz, tgt_len, _ = hidden_states.shape
        source = hidden_states if key_value_states is None else key_value_states
        heads = self.num_heads
        xq, xk, xv = self.q_proj(hidden_states) * self.scaling, self.k_proj(source), self.v_proj(source)
        # self-attention in the plain quantising state: the three head-split sites in one launch (same bits); cross-attention
        # (keys / values of another length) and every other state: site by site
        fused = qkv_heads_fake_quant((self.query_post_act_fake_quantize, self.key_post_act_fake_quantize, self.value_post_act_fake_quantize),
                                     (xq, xk, xv), heads) if xk.shape == xq.shape else None
        if fused is not None:
            q, k, v = fused
        else:
            q = split_heads_fake_quant(self.query_post_act_fake_quantize, xq, heads, observation_mask)
            k = split_heads_fake_quant(self.key_post_act_fake_quantize, xk, heads, observation_mask)
            v = split_heads_fake_quant(self.value_post_act_fake_quantize, xv, heads, observation_mask)
        proj = (bsz * self.num_heads, -1, self.head_dim)
        q, k, v = q.view(*proj), k.view(*proj), v.view(*proj)
        src_len = k.shape[1]
        w = torch.bmm(q, k.transpose(1, 2))
        if attention_mask is not None:
            w = (w.view(bsz, self.num_heads, tgt_len, src_len) + attention_mask).view(bsz * self.num_heads, tgt_len, src_len)
        w = nn.functional.softmax(w, dim=-1)
        probs = nn.functional.dropout(w, p=self.dropout, training=self.training)
        probs = self.attention_probs_post_act_fake_quantize(probs, observation_mask, 2)
        out = merge_heads_fake_quant(self.context_post_act_fake_quantize,
                                     torch.bmm(probs, v).view(bsz, self.num_heads, tgt_len, self.head_dim), observation_mask)
        out = self.out_proj(out)
        if self.qoutput:
            out = self.out_proj_post_act_fake_quantize(out, observation_mask, 1)
        return out


class QuantizedBartEncoderLayer(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend)
        self.qoutput = qoutput
        self.embed_dim = org_module.embed_dim
        self.self_attn = QuantizedBartAttention(org_module.self_attn, w_qconfig, a_qconfig, qoutput=False, backend=backend)
        self.before_self_attn_layer_norm_residual = GammaResidual()
        self.self_attn_layer_norm = QuantizedLayerNorm(org_module.self_attn_layer_norm, w_qconfig, a_qconfig,
                                                       qoutput=True, backend=backend)
        self.dropout = org_module.dropout
        self.fc1 = Quantizer(org_module.fc1, w_qconfig)
        self.activation_fn = org_module.activation_fn
        self.activation_dropout = org_module.activation_dropout
        self.fc1_act_fn_post_act_fake_quantize = Quantizer(None, a_qconfig)
        self.fc2 = Quantizer(org_module.fc2, w_qconfig)
        self.before_final_layer_norm_residual = GammaResidual()
        self.final_layer_norm = QuantizedLayerNorm(org_module.final_layer_norm, w_qconfig, a_qconfig, qoutput=qoutput,
                                                   backend=backend)

    def _drop(self, x, p):
        return nn.functional.dropout(x, p=p, training=self.training)

    def forward(self, hidden_states, attention_mask, observation_mask=None):
        residual = hidden_states
        h = self._drop(self.self_attn(hidden_states, attention_mask=attention_mask, observation_mask=observation_mask),
                       self.dropout)
        h = residual_layernorm(self.before_self_attn_layer_norm_residual, self.self_attn_layer_norm, residual, h, observation_mask)
        residual = h
        if self.training and self.activation_dropout > 0:
            h = self._drop(self.activation_fn(self.fc1(h)), self.activation_dropout)
            h = self.fc1_act_fn_post_act_fake_quantize(h, observation_mask, 1)
        else:
            h = activation_fake_quant(self.activation_fn, self.fc1_act_fn_post_act_fake_quantize, self.fc1(h), observation_mask)
        h = self._drop(self.fc2(h), self.dropout)
        return residual_layernorm(self.before_final_layer_norm_residual, self.final_layer_norm, residual, h, observation_mask)


class QuantizedBartDecoderLayer(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend)
        self.qoutput = qoutput
        self.embed_dim = org_module.embed_dim
        self.self_attn = QuantizedBartAttention(org_module.self_attn, w_qconfig, a_qconfig, qoutput=False, backend=backend)
        self.dropout = org_module.dropout
        self.before_self_attn_layer_norm_residual = GammaResidual()
        self.self_attn_layer_norm = QuantizedLayerNorm(org_module.self_attn_layer_norm, w_qconfig, a_qconfig,
                                                       qoutput=True, backend=backend)
        self.encoder_attn = QuantizedBartAttention(org_module.encoder_attn, w_qconfig, a_qconfig, qoutput=False,
                                                   backend=backend)
        self.before_encoder_attn_layer_norm_residual = GammaResidual()
        self.encoder_attn_layer_norm = QuantizedLayerNorm(org_module.encoder_attn_layer_norm, w_qconfig, a_qconfig,
                                                          qoutput=True, backend=backend)
        self.fc1 = Quantizer(org_module.fc1, w_qconfig)
        self.activation_fn = org_module.activation_fn
        self.activation_dropout = org_module.activation_dropout
        self.fc1_act_fn_post_act_fake_quantize = Quantizer(None, a_qconfig)
        self.fc2 = Quantizer(org_module.fc2, w_qconfig)
        self.before_final_layer_norm_residual = GammaResidual()
        self.final_layer_norm = QuantizedLayerNorm(org_module.final_layer_norm, w_qconfig, a_qconfig, qoutput=qoutput,
                                                   backend=backend)

    def _drop(self, x, p):
        return nn.functional.dropout(x, p=p, training=self.training)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                observation_mask=None):
        residual = hidden_states
        h = self._drop(self.self_attn(hidden_states, attention_mask=attention_mask, observation_mask=observation_mask),
                       self.dropout)
        h = residual_layernorm(self.before_self_attn_layer_norm_residual, self.self_attn_layer_norm, residual, h, observation_mask)
        if encoder_hidden_states is not None:
            residual = h
            h = self._drop(self.encoder_attn(h, key_value_states=encoder_hidden_states,
                                             attention_mask=encoder_attention_mask, observation_mask=observation_mask),
                           self.dropout)
            h = residual_layernorm(self.before_encoder_attn_layer_norm_residual, self.encoder_attn_layer_norm, residual, h, observation_mask)
        residual = h
        if self.training and self.activation_dropout > 0:
            h = self._drop(self.activation_fn(self.fc1(h)), self.activation_dropout)
            h = self.fc1_act_fn_post_act_fake_quantize(h, observation_mask, 1)
        else:
            h = activation_fake_quant(self.activation_fn, self.fc1_act_fn_post_act_fake_quantize, self.fc1(h), observation_mask)
        h = self._drop(self.fc2(h), self.dropout)
        return residual_layernorm(self.before_final_layer_norm_residual, self.final_layer_norm, residual, h, observation_mask)


class _BartStack(QuantizedModule):
    """Shared front end of encoder and decoder: token + position embeddings -> LayerNorm(+quantizer)."""

    layer_cls = None

    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.qoutput = qoutput
        self.config = org_module.config
        self.dropout = org_module.dropout
        self.layerdrop = org_module.layerdrop
        self.padding_idx = org_module.padding_idx
        self.embed_scale = _embed_scale(org_module)
        self.embed_tokens = Quantizer(_plain_embedding(org_module.embed_tokens), w_qconfig)
        self.embed_positions = QuantizedBartLearnedPositionalEmbedding(org_module.embed_positions, w_qconfig, a_qconfig,
                                                                       qoutput=False, backend=backend)
        self.layernorm_embedding = QuantizedLayerNorm(org_module.layernorm_embedding, w_qconfig, a_qconfig, qoutput=True,
                                                      backend=backend)
        n = len(org_module.layers)
        self.layers = nn.ModuleList(
            self.layer_cls(org_module.layers[i], w_qconfig, a_qconfig, qoutput=(True if i != n - 1 else qoutput),
                           backend=backend) for i in range(n))

    def _embed(self, input_ids, observation_mask):
        x = self.embed_tokens(input_ids) * self.embed_scale + self.embed_positions(input_ids.shape)
        x = self.layernorm_embedding(x, observation_mask)
        return nn.functional.dropout(x, p=self.dropout, training=self.training)


class QuantizedBartEncoder(_BartStack):
    layer_cls = QuantizedBartEncoderLayer

    def forward(self, input_ids, attention_mask=None, observation_mask=None):
        h = self._embed(input_ids, observation_mask)
        mask = _expand_mask(attention_mask, h.dtype) if attention_mask is not None else None
        for layer in self.layers:
            h = layer(h, mask, observation_mask=observation_mask)
        return h


class QuantizedBartDecoder(_BartStack):
    layer_cls = QuantizedBartDecoderLayer

    def forward(self, input_ids, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                observation_mask=None):
        bsz, tgt_len = input_ids.shape
        h = self._embed(input_ids, observation_mask)
        mask = _causal_mask(bsz, tgt_len, h.dtype, h.device) if tgt_len > 1 else None
        if attention_mask is not None:
            pad = _expand_mask(attention_mask, h.dtype, tgt_len=tgt_len)
            mask = pad if mask is None else pad + mask
        enc_mask = None
        if encoder_hidden_states is not None and encoder_attention_mask is not None:
            enc_mask = _expand_mask(encoder_attention_mask, h.dtype, tgt_len=tgt_len)
        for layer in self.layers:
            h = layer(h, attention_mask=mask, encoder_hidden_states=encoder_hidden_states,
                      encoder_attention_mask=enc_mask, observation_mask=observation_mask)
        return h


class QuantizedBartModel(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.qoutput = qoutput
        self.config = org_module.config
        self.shared = Quantizer(_plain_embedding(org_module.shared), w_qconfig)
        self.encoder = QuantizedBartEncoder(org_module.encoder, w_qconfig, a_qconfig, qoutput=True, backend=backend)
        self.decoder = QuantizedBartDecoder(org_module.decoder, w_qconfig, a_qconfig, qoutput=qoutput, backend=backend)

    def forward(self, input_ids=None, attention_mask=None, decoder_input_ids=None, decoder_attention_mask=None,
                observation_mask=None, decoder_observation_mask=None):
        if decoder_input_ids is None:
            decoder_input_ids = shift_tokens_right(input_ids, self.config.pad_token_id, self.config.decoder_start_token_id)
        enc = self.encoder(input_ids, attention_mask=attention_mask, observation_mask=observation_mask)
        dec = self.decoder(decoder_input_ids, attention_mask=decoder_attention_mask, encoder_hidden_states=enc,
                           encoder_attention_mask=attention_mask, observation_mask=decoder_observation_mask)
        return dec, enc


class QuantizedBartForConditionalGeneration(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic", is_remove_padding=False):
        super().__init__(backend)
        self.is_remove_padding = is_remove_padding
        self.config = org_module.config
        self.qoutput = qoutput
        self.model = QuantizedBartModel(org_module.model, w_qconfig, a_qconfig, qoutput=True, backend=backend)
        self.lm_head = Quantizer(org_module.lm_head, w_qconfig)
        self.register_buffer("final_logits_bias", org_module.final_logits_bias.clone())

    def forward(self, input_ids=None, attention_mask=None, decoder_input_ids=None, decoder_attention_mask=None,
                labels=None, **unused):
        obs = dec_obs = None
        if self.is_remove_padding:                      # quant_bart.py:1064-1072
            obs = attention_mask.sum(1)
            dec_obs = obs if decoder_attention_mask is None else decoder_attention_mask.sum(1)
        if labels is not None and decoder_input_ids is None:
            decoder_input_ids = shift_tokens_right(labels, self.config.pad_token_id, self.config.decoder_start_token_id)
        dec, enc = self.model(input_ids, attention_mask, decoder_input_ids, decoder_attention_mask,
                              observation_mask=obs, decoder_observation_mask=dec_obs)
        logits = self.lm_head(dec) + self.final_logits_bias
        return with_loss(lm_loss(logits, labels, self.config.vocab_size), (logits, enc))


class QuantizedBartClassificationHead(QuantizedModule):
    """quant_bart.py:505-529: the sentence representation, the tanh layer's output (and the logits) are quantizer sites;
    flat [B, H] tensors -- no mask, no sequence axis."""

    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend)
        self.qoutput = qoutput
        self.getitem_post_act_fake_quantize = Quantizer(None, a_qconfig)
        self.dense = Quantizer(org_module.dense, w_qconfig)
        self.dropout = org_module.dropout
        self.dropout_post_act_fake_quantize = Quantizer(None, a_qconfig)
        self.out_proj = Quantizer(org_module.out_proj, w_qconfig)
        if qoutput:
            self.out_proj_post_act_fake_quantize = Quantizer(None, a_qconfig)

    def forward(self, hidden_states):
        hidden_states = self.getitem_post_act_fake_quantize(self.dropout(hidden_states))
        hidden_states = self.dropout(torch.tanh(self.dense(hidden_states)))
        hidden_states = self.out_proj(self.dropout_post_act_fake_quantize(hidden_states))
        if self.qoutput:
            hidden_states = self.out_proj_post_act_fake_quantize(hidden_states)
        return hidden_states


def _observation_masks(is_remove_padding, attention_mask, decoder_attention_mask):
    """quant_bart.py:1064-1072 / 1205-1213 / 1334-1342."""
    if not is_remove_padding:
        return None, None
    obs = attention_mask.sum(1)
    return obs, (obs if decoder_attention_mask is None else decoder_attention_mask.sum(1))


class QuantizedBartForSequenceClassification(QuantizedModule):
    """quant_bart.py:1167-1289: the decoder state at the last <eos> token through the quantized classification head."""

    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic", is_remove_padding=False):
        super().__init__(backend)
        self.is_remove_padding = is_remove_padding
        self.qoutput = qoutput
        self.config = org_module.config
        self.model = QuantizedBartModel(org_module.model, w_qconfig, a_qconfig, qoutput=False, backend=backend)
        self.classification_head = QuantizedBartClassificationHead(org_module.classification_head, w_qconfig, a_qconfig,
                                                                   qoutput=qoutput, backend=backend)

    def forward(self, input_ids=None, attention_mask=None, decoder_input_ids=None, decoder_attention_mask=None, labels=None,
                **unused):
        obs, dec_obs = _observation_masks(self.is_remove_padding, attention_mask, decoder_attention_mask)
        dec, enc = self.model(input_ids, attention_mask, decoder_input_ids, decoder_attention_mask,
                              observation_mask=obs, decoder_observation_mask=dec_obs)
        eos_mask = input_ids.eq(self.config.eos_token_id)
        if len(torch.unique_consecutive(eos_mask.sum(1))) > 1:
            raise ValueError("All examples must have the same number of <eos> tokens.")
        sentence = dec[eos_mask, :].view(dec.size(0), -1, dec.size(-1))[:, -1, :]
        logits = self.classification_head(sentence)
        return with_loss(classification_loss(self.config, self.config.num_labels, logits, labels), (logits, enc))


class QuantizedBartForQuestionAnswering(QuantizedModule):
    """quant_bart.py:1292-1408: start / end logits from the (quantized) decoder output."""

    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic", is_remove_padding=False):
        super().__init__(backend)
        self.is_remove_padding = is_remove_padding
        self.qoutput = qoutput
        self.num_labels = org_module.num_labels
        self.config = org_module.config
        self.model = QuantizedBartModel(org_module.model, w_qconfig, a_qconfig, qoutput=True, backend=backend)
        self.qa_outputs = Quantizer(org_module.qa_outputs, w_qconfig)
        if qoutput:
            self.qa_outputs_post_act_fake_quantize = Quantizer(None, a_qconfig)

    def forward(self, input_ids=None, attention_mask=None, decoder_input_ids=None, decoder_attention_mask=None,
                start_positions=None, end_positions=None, **unused):
        obs, dec_obs = _observation_masks(self.is_remove_padding, attention_mask, decoder_attention_mask)
        dec, enc = self.model(input_ids, attention_mask, decoder_input_ids, decoder_attention_mask,
                              observation_mask=obs, decoder_observation_mask=dec_obs)
        logits = self.qa_outputs(dec)
        if self.qoutput:
            logits = self.qa_outputs_post_act_fake_quantize(logits)
        start, end = logits.split(1, dim=-1)
        start, end = start.squeeze(-1).contiguous(), end.squeeze(-1).contiguous()
        return with_loss(span_loss(start, end, start_positions, end_positions), (start, end, enc))
