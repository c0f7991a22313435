"""vita/conversation.py of the reference (prompt templates)."""
from vita_amd.host.prompt import (Conversation, SeparatorStyle, conv_mixtral_two, conv_mixtral_zh,  # noqa: F401
                                  conv_templates, default_conversation)
