"""Oracle GMMN generator (zs3/modeling/gmmn.py:6-49).  Test infrastructure only."""
import torch
import torch.nn as nn


class GMMNnetwork(nn.Module):
    def __init__(self, noise_dim, embed_dim, hidden_size, feature_dim, semantic_reconstruction=False):
        super().__init__()
        d_in = noise_dim + embed_dim
        if hidden_size:
            self.model = nn.Sequential(
                nn.Linear(d_in, hidden_size), nn.LeakyReLU(0.2), nn.Dropout(0.5), nn.Linear(hidden_size, feature_dim)
            )
            linears = [self.model[0], self.model[3]]
        else:
            self.model = nn.Linear(d_in, feature_dim)
            linears = [self.model]
        for lin in linears:  # gmmn.py:23-26,36
            nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.fill_(0.01)
        self.semantic_reconstruction = semantic_reconstruction
        if semantic_reconstruction:  # created after the init pass: keeps default init (gmmn.py:37-41)
            self.semantic_reconstruction_layer = nn.Linear(feature_dim, d_in)

    def forward(self, embd, noise):
        feat = self.model(torch.cat((embd, noise), 1))
        if self.semantic_reconstruction:
            return feat, self.semantic_reconstruction_layer(feat)
        return feat
