"""Quantized BERT: where the activation / weight quantizers sit.

Placement table (reference: quant_transformer/model/quant_bert.py):
  embeddings   word/position/token_type -> QEmbedding (:49-51); LayerNorm output, seq axis 1 (:59-60,93)
  self-attn    query/key/value -> QLinear (:109-111); q [B,h,T,d] seq axis 2 (:148); k^T [B,h,d,T] seq axis 3
               (:150); probs [B,h,T,T] seq axis 2 (:185); v seq axis 2 (:186); context [B,T,H] seq axis 1 (:193)
  self-output  dense -> GammaResidual(shortcut, hidden) -> LayerNorm (+quantizer) (:211-216)
  ffn          intermediate.dense -> act -> quantizer seq axis 1 (:277-280); output as self-output (:298-303)
  encoder      last layer's output LayerNorm quantized only if qoutput (:359-362)
  pooler       hidden[:, 0] -> quantizer (no mask) -> dense -> tanh [-> quantizer if qoutput] (:445-451)
  cls head     dropout -> quantizer (no mask) -> classifier (:651-655); observation_mask = attention_mask.sum(1) (:632)
  qa head      bert(qoutput=True, no pooler) -> qa_outputs (:698-703,744-747)
Sub-modules are registered in the same order as there: Gamma Migration pairs a LayerNorm with the
NEXT GammaResidual in named_modules() order (gamma_migration.py:58-75).
"""
import math

import torch
from torch import nn

from .losses import classification_loss, span_loss, with_loss

from ..quantization import QuantizedModule, Quantizer
from ..util_layernorm import (GammaResidual, QuantizedLayerNorm, activation_fake_quant, merge_heads_fake_quant,
                              qkv_heads_fake_quant, residual_layernorm)


class QuantizedBertEmbeddings(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.qoutput = qoutput
        self.word_embeddings = Quantizer(org_module.word_embeddings, w_qconfig)
        self.position_embeddings = Quantizer(org_module.position_embeddings, w_qconfig)
        self.token_type_embeddings = Quantizer(org_module.token_type_embeddings, w_qconfig)
        self.dropout = org_module.dropout
        self.position_embedding_type = getattr(org_module, "position_embedding_type", "absolute")
        self.register_buffer("position_ids", org_module.position_ids.clone())
        self.LayerNorm = QuantizedLayerNorm(org_module.LayerNorm, w_qconfig, a_qconfig, qoutput=qoutput, backend=backend)

    def forward(self, input_ids, token_type_ids=None, position_ids=None, observation_mask=None):
        seq_len = input_ids.shape[1]
        if position_ids is None:
            position_ids = self.position_ids[:, :seq_len]
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        x = self.word_embeddings(input_ids) + self.token_type_embeddings(token_type_ids)
        if self.position_embedding_type == "absolute":
            x += self.position_embeddings(position_ids)
        return self.dropout(self.LayerNorm(x, observation_mask))


class QuantizedBertSelfAttention(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.qoutput = qoutput
        self.num_attention_heads = org_module.num_attention_heads
        self.attention_head_size = org_module.attention_head_size
        self.all_head_size = org_module.all_head_size
        self.query = Quantizer(org_module.query, w_qconfig)
        self.key = Quantizer(org_module.key, w_qconfig)
        self.value = Quantizer(org_module.value, w_qconfig)
        self.dropout = org_module.dropout
        self.query_permute_post_act_fake_quantize = Quantizer(None, a_qconfig)
        self.key_transpose_post_act_fake_quantize = Quantizer(None, a_qconfig)
        self.value_permute_post_act_fake_quantize = Quantizer(None, a_qconfig)
        self.attention_probs_post_act_fake_quantize = Quantizer(None, a_qconfig)
        if qoutput:
            self.context_view_post_act_fake_quantize = Quantizer(None, a_qconfig)

    def _heads(self, x):       # [B,T,H] -> [B,h,T,d] as a VIEW of the [B,T,H] memory
        b, t, _ = x.shape
        return x.view(b, t, self.num_attention_heads, self.attention_head_size).permute(0, 2, 1, 3)

    def forward(self, hidden_states, attention_mask=None, observation_mask=None):
        xq, xk, xv = self.query(hidden_states), self.key(hidden_states), self.value(hidden_states)
        # plain quantising state: the three head-split sites in one launch (same bits as the three calls below)
        fused = qkv_heads_fake_quant((self.query_permute_post_act_fake_quantize, self.key_transpose_post_act_fake_quantize,
                                      self.value_permute_post_act_fake_quantize), (xq, xk, xv), self.num_attention_heads)
        if fused is not None:
            q, kt, v = fused[0], fused[1].transpose(-1, -2), fused[2]
        else:
            q, k, v = self._heads(xq), self._heads(xk), self._heads(xv)
            q = self.query_permute_post_act_fake_quantize(q, observation_mask, 2)
            kt = self.key_transpose_post_act_fake_quantize(k.transpose(-1, -2), observation_mask, 3)
        scores = torch.matmul(q, kt)
        root = math.sqrt(self.attention_head_size)
        if attention_mask is not None and root == 2.0 ** round(math.log2(root)) and not torch.is_grad_enabled():
            # head sizes 16 / 64 / 256: dividing by sqrt(d) is an exact multiplication by a power of two, so
            # mask + scores * (1/sqrt(d)) in ONE stock kernel has the bits of the reference's two (quant_bert.py:172-176)
            scores = torch.add(attention_mask, scores, alpha=1.0 / root)
        else:
            scores = scores / root
            if attention_mask is not None:
                scores = scores + attention_mask
        probs = self.dropout(nn.functional.softmax(scores, dim=-1))
        probs = self.attention_probs_post_act_fake_quantize(probs, observation_mask, 2)
        if fused is None:
            v = self.value_permute_post_act_fake_quantize(v, observation_mask, 2)
        return merge_heads_fake_quant(self.context_view_post_act_fake_quantize if self.qoutput else None,
                                      torch.matmul(probs, v), observation_mask)


class _DenseResidualNorm(QuantizedModule):
    """dense -> dropout -> GammaResidual(shortcut, hidden) -> LayerNorm(+quantizer); BertSelfOutput and BertOutput."""

    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend)
        self.qoutput = qoutput
        self.dense = Quantizer(org_module.dense, w_qconfig)
        self.dropout = org_module.dropout
        self.before_LayerNorm_residual = GammaResidual()
        self.LayerNorm = QuantizedLayerNorm(org_module.LayerNorm, w_qconfig, a_qconfig, qoutput=qoutput, backend=backend)

    def forward(self, hidden_states, input_tensor, observation_mask=None):
        hidden_states = self.dropout(self.dense(hidden_states))
        return residual_layernorm(self.before_LayerNorm_residual, self.LayerNorm, input_tensor, hidden_states, observation_mask)


class QuantizedBertSelfOutput(_DenseResidualNorm):
    pass


class QuantizedBertOutput(_DenseResidualNorm):
    pass


class QuantizedBertAttention(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend)
        self.qoutput = qoutput
        self.self = QuantizedBertSelfAttention(org_module.self, w_qconfig, a_qconfig, qoutput=True, backend=backend)
        self.output = QuantizedBertSelfOutput(org_module.output, w_qconfig, a_qconfig, qoutput=qoutput, backend=backend)

    def forward(self, hidden_states, attention_mask=None, observation_mask=None):
        ctx = self.self(hidden_states, attention_mask, observation_mask=observation_mask)
        return self.output(ctx, hidden_states, observation_mask=observation_mask)


class QuantizedBertIntermediate(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend)
        self.qoutput = qoutput
        self.dense = Quantizer(org_module.dense, w_qconfig)
        self.intermediate_act_fn = org_module.intermediate_act_fn
        if qoutput:
            self.intermediate_act_fn_post_act_fake_quantize = Quantizer(None, a_qconfig)

    def forward(self, hidden_states, observation_mask=None):
        return activation_fake_quant(self.intermediate_act_fn,
                                     self.intermediate_act_fn_post_act_fake_quantize if self.qoutput else None,
                                     self.dense(hidden_states), observation_mask)


class QuantizedBertLayer(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.qoutput = qoutput
        self.attention = QuantizedBertAttention(org_module.attention, w_qconfig, a_qconfig, qoutput=True, backend=backend)
        self.intermediate = QuantizedBertIntermediate(org_module.intermediate, w_qconfig, a_qconfig, qoutput=True,
                                                      backend=backend)
        self.output = QuantizedBertOutput(org_module.output, w_qconfig, a_qconfig, qoutput=qoutput, backend=backend)

    def forward(self, hidden_states, attention_mask=None, observation_mask=None):
        att = self.attention(hidden_states, attention_mask, observation_mask=observation_mask)
        inter = self.intermediate(att, observation_mask=observation_mask)
        return self.output(inter, att, observation_mask=observation_mask)


class QuantizedBertEncoder(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.qoutput = qoutput
        n = len(org_module.layer)
        self.layer = nn.ModuleList(
            QuantizedBertLayer(org_module.layer[i], w_qconfig, a_qconfig, qoutput=(True if i != n - 1 else qoutput),
                               backend=backend) for i in range(n))

    def forward(self, hidden_states, attention_mask=None, observation_mask=None):
        for layer in self.layer:
            hidden_states = layer(hidden_states, attention_mask, observation_mask=observation_mask)
        return hidden_states


class QuantizedBertPooler(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.qoutput = qoutput
        self.getitem_post_act_fake_quantize = Quantizer(None, a_qconfig)
        self.dense = Quantizer(org_module.dense, w_qconfig)
        self.activation = org_module.activation
        if qoutput:
            self.pooler_post_act_fake_quantize = Quantizer(None, a_qconfig)

    def forward(self, hidden_states):
        first = self.getitem_post_act_fake_quantize(hidden_states[:, 0])      # no mask, no seq axis
        pooled = self.activation(self.dense(first))
        if self.qoutput:
            pooled = self.pooler_post_act_fake_quantize(pooled)
        return pooled


class QuantizedBertModel(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.config = org_module.config
        self.qoutput = qoutput
        self.embeddings = QuantizedBertEmbeddings(org_module.embeddings, w_qconfig, a_qconfig, qoutput=True, backend=backend)
        has_pooler = getattr(org_module, "pooler", None) is not None
        self.encoder = QuantizedBertEncoder(org_module.encoder, w_qconfig, a_qconfig,
                                            qoutput=(False if has_pooler else qoutput), backend=backend)
        self.pooler = (QuantizedBertPooler(org_module.pooler, w_qconfig, a_qconfig, qoutput=qoutput, backend=backend)
                       if has_pooler else None)

    def forward(self, input_ids, attention_mask=None, token_type_ids=None, position_ids=None, observation_mask=None):
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        # additive mask with the transformers-4.18 constant the reference was run with
        extended = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -10000.0
        x = self.embeddings(input_ids, token_type_ids, position_ids, observation_mask=observation_mask)
        seq = self.encoder(x, extended, observation_mask=observation_mask)
        pooled = self.pooler(seq) if self.pooler is not None else None
        return seq, pooled


def _observation_mask(attention_mask, enabled):
    """quant_bert.py:632-635: valid length per sample (right-padded batches)."""
    return attention_mask.sum(1) if (enabled and attention_mask is not None) else None


class QuantizedBertForSequenceClassification(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic", is_remove_padding=False):
        super().__init__(backend)
        self.is_remove_padding = is_remove_padding
        self.num_labels = org_module.num_labels
        self.config = org_module.config
        self.qoutput = qoutput
        self.bert = QuantizedBertModel(org_module.bert, w_qconfig, a_qconfig, qoutput=False, backend=backend)
        self.dropout = org_module.dropout
        self.dropout_post_act_fake_quantize = Quantizer(None, a_qconfig)
        self.classifier = Quantizer(org_module.classifier, w_qconfig)
        if qoutput:
            self.classifier_post_act_fake_quantize = Quantizer(None, a_qconfig)

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, labels=None, **unused):
        obs = _observation_mask(attention_mask, self.is_remove_padding)
        _, pooled = self.bert(input_ids, attention_mask, token_type_ids, position_ids, observation_mask=obs)
        logits = self.classifier(self.dropout_post_act_fake_quantize(self.dropout(pooled)))
        if self.qoutput:
            logits = self.classifier_post_act_fake_quantize(logits)
        return with_loss(classification_loss(self.config, self.num_labels, logits, labels), (logits,))


class QuantizedBertForQuestionAnswering(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic", is_remove_padding=False):
        super().__init__(backend)
        self.is_remove_padding = is_remove_padding
        self.config = org_module.config
        self.num_labels = org_module.num_labels
        self.bert = QuantizedBertModel(org_module.bert, w_qconfig, a_qconfig, qoutput=True, backend=backend)
        self.qa_outputs = Quantizer(org_module.qa_outputs, w_qconfig)
        self.qoutput = qoutput
        if qoutput:
            self.qa_outputs_post_act_fake_quantize = Quantizer(None, a_qconfig)

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, start_positions=None,
                end_positions=None, **unused):
        obs = _observation_mask(attention_mask, self.is_remove_padding)
        seq, _ = self.bert(input_ids, attention_mask, token_type_ids, position_ids, observation_mask=obs)
        logits = self.qa_outputs(seq)
        if self.qoutput:
            logits = self.qa_outputs_post_act_fake_quantize(logits)
        start, end = logits.split(1, dim=-1)
        start, end = start.squeeze(-1).contiguous(), end.squeeze(-1).contiguous()
        return with_loss(span_loss(start, end, start_positions, end_positions), (start, end))
