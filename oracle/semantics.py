"""oracle/semantics.py — pure-Python restatement of go-ibft's message store and quorum
rule (TEST INFRASTRUCTURE ONLY; small cases).

Written line-by-line from the Go sources with Python dicts standing in for Go maps:
  Messages              /root/reference/messages/messages.go:25-323
  ValidatorManager      /root/reference/core/validator_manager.go:22-155
Messages are ``oracle.wire.IbftMessage`` objects.  Pinned by replaying the reference's
own unit tables (tests/test_host_semantics.py), then used as the oracle for randomized
comparisons against the C++ host mirror.
"""
from __future__ import annotations


class Messages:
    def __init__(self):
        self.maps = {t: {} for t in range(4)}          # type -> height -> round -> from -> msg

    def add_message(self, m):                           # messages.go:54-65
        self.maps[m.type].setdefault(m.view.height, {}).setdefault(m.view.round, {})[bytes(m.sender)] = m

    def num_messages(self, height, round_, type_):      # :96-119
        return len(self.maps[type_].get(height, {}).get(round_, {}))

    def prune_by_height(self, height):                  # :123-148
        for t in range(4):
            for h in [h for h in self.maps[t] if h < height]:
                del self.maps[t][h]

    def get_valid_messages(self, height, round_, type_, is_valid):   # :169-199
        msgs = self.maps[type_].get(height, {}).get(round_)
        if msgs is None:
            return []
        valid, invalid = [], []
        for key, m in msgs.items():
            if not is_valid(m):
                invalid.append(key)
                continue
            valid.append(m)
        for key in invalid:
            del msgs[key]
        return valid

    def get_extended_rcc(self, height, is_valid_message, is_valid_rcc):   # :202-245
        round_map = self.maps[3].get(height, {})
        highest, extended = 0, []
        for round_, msgs in round_map.items():
            if round_ <= highest:
                continue
            valid = [m for m in msgs.values() if is_valid_message(m)]
            if not is_valid_rcc(round_, valid):
                continue
            highest, extended = round_, valid
        return extended

    def get_most_round_change_messages(self, min_round, height):          # :249-286
        round_map = self.maps[3].get(height, {})
        best, best_count = 0, 0
        for round_, msgs in round_map.items():
            if round_ < min_round:
                continue
            if len(msgs) > best_count:
                best, best_count = round_, len(msgs)
        if best == 0:
            return []
        return list(round_map[best].values())


class ValidatorManager:
    def __init__(self):
        self.power = None
        self.quorum = 0

    def init(self, powers: dict) -> bool:               # setCurrentVotingPower :61-75
        total = sum(powers.values())
        if total <= 0:
            return False
        self.power = dict(powers)
        self.quorum = 2 * total // 3 + 1                 # calculateQuorum :130-135
        return True

    def has_quorum(self, senders) -> bool:              # :77-96
        if self.power is None:
            return False
        return sum(self.power.get(bytes(s), 0) for s in set(map(bytes, senders))) >= self.quorum

    def has_prepare_quorum(self, proposal_message, msgs) -> bool:   # :99-127
        if proposal_message is None:
            return False
        senders = {bytes(proposal_message.sender)}
        for m in msgs:
            if bytes(m.sender) == bytes(proposal_message.sender):
                return False
            senders.add(bytes(m.sender))
        return self.has_quorum(senders)
