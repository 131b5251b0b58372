class FusedMoE:
    def forward(self, hidden_states, router_logits):
        return ("custom_op", self.forward_impl(hidden_states, router_logits))

    def forward_impl(self, hidden_states, router_logits):
        return hidden_states * 2
