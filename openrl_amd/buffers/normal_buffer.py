"""``NormalReplayBuffer`` - thin holder of a device ``ReplayData`` (``openrl/buffers/normal_buffer.py:22-108``)."""
from .replay_data import ReplayData


class NormalReplayBuffer(object):
    def __init__(self, cfg, num_agents, obs_space, act_space, data_client=None, episode_length=None, device=None):
        self.data = ReplayData(cfg, num_agents, obs_space, act_space, data_client, episode_length, device=device)

    def init_buffer(self, raw_obs, action_masks=None):
        self.data.init_buffer(raw_obs, action_masks)

    def insert(self, raw_obs, rnn_states, rnn_states_critic, actions, action_log_probs, value_preds, rewards, masks,
               bad_masks=None, active_masks=None, action_masks=None):
        self.data.insert(raw_obs, rnn_states, rnn_states_critic, actions, action_log_probs, value_preds, rewards, masks,
                         bad_masks, active_masks, action_masks)

    def after_update(self):
        self.data.after_update()

    def compute_returns(self, next_value, value_normalizer=None):
        self.data.compute_returns(next_value, value_normalizer)

    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None,
                               critic_obs_process_func=None):
        return self.data.feed_forward_generator(advantages, num_mini_batch, mini_batch_size,
                                                critic_obs_process_func=critic_obs_process_func)
