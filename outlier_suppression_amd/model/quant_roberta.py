"""Quantized RoBERTa: BERT's encoder placement plus RoBERTa's position ids and classification head.

Reference: quant_transformer/model/quant_roberta.py.  Differences from BERT that matter here:
  embeddings  position ids count non-padding tokens from padding_idx+1 (:825-838, fairseq make_positions)
  cls model   RobertaModel is built without a pooler, so the encoder is constructed with the model's own
              ``qoutput`` (False for the task models: the last output LayerNorm stays unquantized, :497-501)
  cls head    features[:, 0] -> dropout -> quantizer -> dense -> tanh -> dropout -> quantizer -> out_proj (:621-644)
Sub-module names follow the reference (``roberta.*``, ``classifier.dense`` ...), so Gamma Migration's
``get_weight_modules`` and the name-based switches work unchanged.
"""
import torch

from .losses import classification_loss, span_loss, with_loss
from ..quantization import QuantizedModule, Quantizer
from ..util_layernorm import QuantizedLayerNorm
from . import quant_bert as B


def create_position_ids_from_input_ids(input_ids, padding_idx, past_key_values_length=0):
    """quant_roberta.py:825-838: padding keeps padding_idx, real tokens get padding_idx + their 1-based rank."""
    mask = input_ids.ne(padding_idx).int()
    return ((torch.cumsum(mask, dim=1).type_as(mask) + past_key_values_length) * mask).long() + padding_idx


class QuantizedRobertaEmbeddings(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.qoutput = qoutput
        self.word_embeddings = Quantizer(org_module.word_embeddings, w_qconfig)
        self.position_embeddings = Quantizer(org_module.position_embeddings, w_qconfig)
        self.padding_idx = org_module.padding_idx
        self.token_type_embeddings = Quantizer(org_module.token_type_embeddings, w_qconfig)
        self.dropout = org_module.dropout
        self.position_embedding_type = getattr(org_module, "position_embedding_type", "absolute")
        self.register_buffer("position_ids", org_module.position_ids.clone())
        self.LayerNorm = QuantizedLayerNorm(org_module.LayerNorm, w_qconfig, a_qconfig, qoutput=qoutput, backend=backend)

    def forward(self, input_ids, token_type_ids=None, position_ids=None, observation_mask=None):
        if position_ids is None:
            position_ids = create_position_ids_from_input_ids(input_ids, self.padding_idx)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        x = self.word_embeddings(input_ids) + self.token_type_embeddings(token_type_ids)
        if self.position_embedding_type == "absolute":
            x += self.position_embeddings(position_ids)
        return self.dropout(self.LayerNorm(x, observation_mask))


class QuantizedRobertaSelfAttention(B.QuantizedBertSelfAttention):
    pass


class QuantizedRobertaSelfOutput(B.QuantizedBertSelfOutput):
    pass


class QuantizedRobertaIntermediate(B.QuantizedBertIntermediate):
    pass


class QuantizedRobertaOutput(B.QuantizedBertOutput):
    pass


class QuantizedRobertaPooler(B.QuantizedBertPooler):
    pass


class QuantizedRobertaAttention(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend)
        self.qoutput = qoutput
        self.self = QuantizedRobertaSelfAttention(org_module.self, w_qconfig, a_qconfig, qoutput=True, backend=backend)
        self.output = QuantizedRobertaSelfOutput(org_module.output, w_qconfig, a_qconfig, qoutput=qoutput, backend=backend)

    def forward(self, hidden_states, attention_mask=None, observation_mask=None):
        ctx = self.self(hidden_states, attention_mask, observation_mask=observation_mask)
        return self.output(ctx, hidden_states, observation_mask=observation_mask)


class QuantizedRobertaLayer(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.qoutput = qoutput
        self.attention = QuantizedRobertaAttention(org_module.attention, w_qconfig, a_qconfig, qoutput=True, backend=backend)
        self.intermediate = QuantizedRobertaIntermediate(org_module.intermediate, w_qconfig, a_qconfig, qoutput=True,
                                                         backend=backend)
        self.output = QuantizedRobertaOutput(org_module.output, w_qconfig, a_qconfig, qoutput=qoutput, backend=backend)

    def forward(self, hidden_states, attention_mask=None, observation_mask=None):
        att = self.attention(hidden_states, attention_mask, observation_mask=observation_mask)
        inter = self.intermediate(att, observation_mask=observation_mask)
        return self.output(inter, att, observation_mask=observation_mask)


class QuantizedRobertaEncoder(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.qoutput = qoutput
        n = len(org_module.layer)
        self.layer = torch.nn.ModuleList(
            QuantizedRobertaLayer(org_module.layer[i], w_qconfig, a_qconfig, qoutput=(True if i != n - 1 else qoutput),
                                  backend=backend) for i in range(n))

    def forward(self, hidden_states, attention_mask=None, observation_mask=None):
        for layer in self.layer:
            hidden_states = layer(hidden_states, attention_mask, observation_mask=observation_mask)
        return hidden_states


class QuantizedRobertaModel(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.config = org_module.config
        self.qoutput = qoutput
        self.embeddings = QuantizedRobertaEmbeddings(org_module.embeddings, w_qconfig, a_qconfig, qoutput=True,
                                                     backend=backend)
        has_pooler = getattr(org_module, "pooler", None) is not None
        self.encoder = QuantizedRobertaEncoder(org_module.encoder, w_qconfig, a_qconfig,
                                               qoutput=(False if has_pooler else qoutput), backend=backend)
        self.pooler = (QuantizedRobertaPooler(org_module.pooler, w_qconfig, a_qconfig, qoutput=qoutput, backend=backend)
                       if has_pooler else None)

    def forward(self, input_ids, attention_mask=None, token_type_ids=None, position_ids=None, observation_mask=None):
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        extended = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -10000.0
        x = self.embeddings(input_ids, token_type_ids, position_ids, observation_mask=observation_mask)
        seq = self.encoder(x, extended, observation_mask=observation_mask)
        pooled = self.pooler(seq) if self.pooler is not None else None
        return seq, pooled


class QuantizedRobertaClassificationHead(QuantizedModule):
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

    def forward(self, features, **kwargs):
        x = self.getitem_post_act_fake_quantize(self.dropout(features[:, 0, :]))      # <s> token, no mask
        x = self.dropout(torch.tanh(self.dense(x)))
        x = self.out_proj(self.dropout_post_act_fake_quantize(x))
        if self.qoutput:
            x = self.out_proj_post_act_fake_quantize(x)
        return x


class QuantizedRobertaForSequenceClassification(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic", is_remove_padding=False):
        super().__init__(backend)
        self.is_remove_padding = is_remove_padding
        self.num_labels = org_module.num_labels
        self.config = org_module.config
        self.qoutput = qoutput
        self.roberta = QuantizedRobertaModel(org_module.roberta, w_qconfig, a_qconfig, qoutput=False, backend=backend)
        self.classifier = QuantizedRobertaClassificationHead(org_module.classifier, w_qconfig, a_qconfig, qoutput=qoutput,
                                                             backend=backend)

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, labels=None, **unused):
        obs = B._observation_mask(attention_mask, self.is_remove_padding)
        seq, _ = self.roberta(input_ids, attention_mask, token_type_ids, position_ids, observation_mask=obs)
        logits = self.classifier(seq)
        return with_loss(classification_loss(self.config, self.num_labels, logits, labels), (logits,))


class QuantizedRobertaForQuestionAnswering(QuantizedModule):
    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic", is_remove_padding=False):
        super().__init__(backend)
        self.is_remove_padding = is_remove_padding
        self.config = org_module.config
        self.qoutput = qoutput
        self.num_labels = org_module.config.num_labels
        self.roberta = QuantizedRobertaModel(org_module.roberta, w_qconfig, a_qconfig, qoutput=True, backend=backend)
        self.qa_outputs = Quantizer(org_module.qa_outputs, w_qconfig)
        if qoutput:
            self.qa_outputs_post_act_fake_quantize = Quantizer(None, a_qconfig)

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, start_positions=None,
                end_positions=None, **unused):
        obs = B._observation_mask(attention_mask, self.is_remove_padding)
        seq, _ = self.roberta(input_ids, attention_mask, token_type_ids, position_ids, observation_mask=obs)
        logits = self.qa_outputs(seq)
        if self.qoutput:
            logits = self.qa_outputs_post_act_fake_quantize(logits)
        start, end = logits.split(1, dim=-1)
        start, end = start.squeeze(-1).contiguous(), end.squeeze(-1).contiguous()
        return with_loss(span_loss(start, end, start_positions, end_positions), (start, end))
