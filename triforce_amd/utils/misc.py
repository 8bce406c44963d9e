"""Console / CSV helpers (role of the reference's utils/misc.py: spec_stream :5-16, log_csv :23-35,
print_config :37-50).  ANSI colours directly — no termcolor dependency."""
_ANSI = {"red": 31, "green": 32, "yellow": 33, "blue": 34, "magenta": 35, "cyan": 36}


def colored(text, color=None):
    code = _ANSI.get(color)
    return f"\033[{code}m{text}\033[0m" if code else str(text)


def spec_stream(pred_token_idx, tokenizer, color="blue"):
    if tokenizer is None:
        return
    decoded = tokenizer.decode(pred_token_idx, skip_special_tokens=True, clean_up_tokenization_spaces=True)
    print(colored(decoded.replace("<0x0A>", "\n"), color), flush=True, end=" ")


def log_csv(file_path, header, entry):
    """Append ``entry``; a new (or empty) file gets ``header`` first."""
    import os
    fresh = not os.path.exists(file_path) or os.path.getsize(file_path) == 0
    with open(file_path, "a") as f:
        f.write((header if fresh else "") + entry)


def print_config(draft, target, prefill, gen_len, gamma, top_k, top_p, temperature, file_path, method,
                 spec_args=None, dataset=None):
    """The banner of the reference scripts, line for line (harnesses grep it)."""
    fields = {"Dataset": dataset, "Spec Args": spec_args, "Draft": draft.config._name_or_path,
              "Target": target.config._name_or_path, "Prefill Length": prefill, "Generation Length": gen_len,
              "Gamma": gamma, "Sampling Method": f"top_k = {top_k}, top_p = {top_p}, temperature = {temperature}",
              "Log CSV": file_path}
    rule = "#" * 39
    print(colored(f"{rule} Config {rule}", "blue"), flush=True)
    print(colored(f"Method: {method}", "red"), flush=True)
    for key, value in fields.items():
        print(colored(f"{key}: {value}", "blue"), flush=True)
    print(colored("#" * 86 + "\n", "blue"), flush=True)
